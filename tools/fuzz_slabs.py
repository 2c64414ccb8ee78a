"""One-off fuzz of the slab protocol on one GPU: random particle counts, world sizes 2..6, jitter and velocities
(below one column per step), re-balancing and outer-edge trimming every few steps — checks conservation, the
violation counters, and statistics against the single-domain engine.  python tools/fuzz_slabs.py [first] [cases] [edge|serial|strips]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import gpu_fluid_simulation_amd as g
from test_multi_gpu import InProcessSlabs
from tests.slab_oracle import match_and_compare
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
step_mode = sys.argv[3] if len(sys.argv) > 3 else "edge"      # edge (the default step) | serial | strips
t0 = time.time()
for case in range(first, first + cases):
    rng = np.random.default_rng(7000 + case)
    side = int(rng.integers(48, 260))
    n = side * side
    world = int(rng.integers(2, 7))
    st, off, tick = g.dam_break_2d(n)
    margin = int(rng.choice([0, 8, 24]))
    every = int(rng.choice([2, 4]))
    sort_mode = g.FS_SORT_BITONIC if case % 3 == 0 else None        # None: the slab default (counting sort)
    slabs = InProcessSlabs(g, st, off, world, cap=n + 4 * 4096, recv=4096, seed=case, vel=float(rng.choice([0.0, 1.0, 5.0])),
                           trim_margin=margin, sort_mode=sort_mode, serial=step_mode == "serial", strips=step_mode == "strips")
    single = g.FluidSimulation(st, device=0, initial_offset=off, ref_quirks=False)
    single.upload_particles(slabs.initial)
    steps = int(rng.integers(20, 60))
    too_fast = False
    for s in range(1, steps + 1):
        slabs.step(tick); single.tick(tick)
        try:
            if s % every == 0:
                slabs.rebalance(2)
            else:
                slabs.assert_clean()
        except AssertionError:
            # The scenario promises speeds below one column per step (see above), and the fixed boundary zone of the overlapped
            # steps (4 columns, never re-sized here) is good for that.  Two particles the jitter put almost on top of each
            # other break the promise (case 44: |v| = 75 after four steps = 3 columns per step): the engine must COUNT that
            # (far_halo / lost) — which is what just happened — and the production driver raises on it.  Not a failure of the case.
            vmax = max(float(np.abs(x.download()[0]["velocity"]).max()) for x in slabs.sims)
            if vmax * tick.delta / st.smoothing_radius < 1.0:
                raise
            print(f"case {case}: a particle reached |v| = {vmax:.0f} ({vmax * tick.delta / st.smoothing_radius:.1f} columns per step) at step {s}: "
                  f"counted by the engine ({[x.counters() for x in slabs.sims if x.counters()['far_halo'] or x.counters()['lost']][:1]}), case cut short", flush=True)
            too_fast = True
            break
        if s == 2:      # elementwise while ULP-level differences (x2.4 per step, faster with random velocities) are still small
            match_and_compare(slabs.owned(), single.download_particles(), st.smoothing_radius, max_key_flips=0.02)
    if too_fast:
        for x in slabs.sims: x.close()
        single.close()
        continue
    slabs.assert_clean()
    own = slabs.owned()
    assert own.shape[0] == n, (case, own.shape[0], n)
    ref = single.download_particles()
    # order-independent statistics.  Not the maximum speed: with random initial velocities the run is chaotic
    # (ULP differences grow ~2.4x per step) and after 20-60 steps an extreme value of ONE particle
    # differs between two equally valid summation orders; bulk statistics and high quantiles do not.
    assert np.isfinite(own["position"]).all() and np.isfinite(own["velocity"]).all()
    np.testing.assert_allclose(own["density"].mean(), ref["density"].mean(), rtol=2e-3)
    np.testing.assert_allclose(own["position"].mean(axis=0), ref["position"].mean(axis=0), atol=2e-3)
    np.testing.assert_allclose(own["velocity"].mean(axis=0), ref["velocity"].mean(axis=0), atol=2e-2)
    sp_a, sp_b = np.hypot(*own["velocity"].T), np.hypot(*ref["velocity"].T)
    for q in (50, 90, 99):
        np.testing.assert_allclose(np.percentile(sp_a, q), np.percentile(sp_b, q), rtol=0.05, atol=1e-3)
    for x in slabs.sims: x.close()
    single.close()
    if (case - first) % 5 == 4:
        print(f"cases {first}..{case} ok (last: n={n}, world {world}, margin {margin}, {steps} steps, columns {np.diff(slabs.bounds).tolist()}) {time.time()-t0:.0f}s", flush=True)
print("slab fuzz ok:", cases, "cases")
