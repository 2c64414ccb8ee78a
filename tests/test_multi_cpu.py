"""N>1 path on CPU: pure partition logic + a world_size-2/3/4 gloo runs of multi.SlabDriver /
multi.Transport with the oracle-backed engine, compared with the single-domain oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_equal_counts(fs):
    from gpu_fluid_simulation_amd import multi
    st, off, tick = fs.dam_break_2d(4096)
    hist, gw = multi.lattice_histogram(fs, st, off)
    assert gw == 66 and hist.sum() == 4096
    for world in (1, 2, 4, 8):
        b = multi.partition_columns(hist, world)
        assert b[0] == 0 and b[-1] == gw and all(b[k + 1] - b[k] >= 4 for k in range(world))
        counts = [hist[b[k]:b[k + 1]].sum() for k in range(world)]
        assert sum(counts) == 4096
        assert max(counts) <= 4096 / world + 2 * hist.max()


def test_trim_outer_edges(fs):
    from gpu_fluid_simulation_amd import multi
    hist = np.zeros(100, dtype=np.int64)
    hist[20:40] = 7
    b = multi.partition_columns(hist, 4)
    t = multi.trim_outer_edges(b, hist, 5)
    assert t[0] == 15 and t[-1] == 45 and t[1:-1] == b[1:-1]
    assert multi.trim_outer_edges(b, hist, 0) == b                       # margin 0: the walls
    assert multi.trim_outer_edges(b, hist, 1000) == b                    # never past the walls
    wide = multi.trim_outer_edges([0, 10, 30, 100], hist, 0)
    assert wide == [0, 10, 30, 100]
    squeezed = multi.trim_outer_edges([0, 21, 39, 100], hist, 0 + 1)     # never inside a neighbour (min 4 columns)
    assert squeezed[0] <= 21 - 4 and squeezed[-1] >= 39 + 4
    assert multi.trim_outer_edges(b, np.zeros(100, dtype=np.int64), 5) == b   # nothing to follow


def test_rebalance_moves_towards_ideal(fs):
    from gpu_fluid_simulation_amd import multi
    hist = np.zeros(100, dtype=np.int64)
    hist[10:60] = 100
    b = [0, 50, 100]                      # ideal boundary is column 35
    nb = multi.rebalance_boundaries(b, hist, max_shift=3)
    assert nb == [0, 47, 100]
    for _ in range(10):
        nb = multi.rebalance_boundaries(nb, hist, max_shift=3)
    assert nb[1] == 35


def test_global_columns_match_oracle_keys(fs, orc):
    from gpu_fluid_simulation_amd import multi
    st, off, tick = fs.dam_break_2d(4096)
    o = orc.OracleSim(st, off)
    o.step(tick)
    p = o.particles()
    gw, gh = o.grid_dims
    cols = multi.global_columns(p["predicted_position"][:, 0], st.size.x, st.smoothing_radius)
    assert np.array_equal(cols, (p["grid"] % gw).astype(np.int64))


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import multi
from oracle import oracle as O
from tests.slab_oracle import OracleSlabEngine
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n, steps = int(sys.argv[2]), int(sys.argv[3])
st, off, tick = g.dam_break_2d(n)
hist, gw = multi.lattice_histogram(g, st, off)
bounds = multi.partition_columns(hist, world)
tr = multi.Transport(rank, world, multi.HEADER_BYTES + multi.RECORD_BYTES * 4096)
eng = OracleSlabEngine(O, st, bounds, rank, world, tr)
eng.upload_owned(multi.initial_owned(g, st, off, bounds, rank))
drv = multi.SlabDriver(eng, tr, bounds, gw, rebalance_every=4, max_shift=1)
for k in range(steps):
    drv.step(tick)
    if k == 1:
        np.save(os.path.join(sys.argv[4], f"early_{rank}.npy"), eng.owned_particles())
own = eng.owned_particles()
np.save(os.path.join(sys.argv[4], f"owned_{rank}.npy"), own)
for _ in range(int(sys.argv[5])):
    drv.step(tick)
np.save(os.path.join(sys.argv[4], f"late_{rank}.npy"), eng.owned_particles())
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3, 4])
def test_gloo_world_slabs_match_single_domain(fs, orc, tmp_path, world):
    n, steps, more = 1024, 5, 11
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + world), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(n), str(steps), str(tmp_path), str(more)],
                              env=dict(env, RANK=str(r)), cwd=ROOT) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.concatenate([np.load(tmp_path / f"owned_{r}.npy") for r in range(world)])
    st, off, tick = fs.dam_break_2d(n)
    ref = orc.OracleSim(st, off, ref_quirks=False)
    from tests.slab_oracle import assert_statistics_close, match_and_compare
    for k in range(steps):
        ref.step(tick)
        if k == 1:    # step 2: cell keys of matched particles are IDENTICAL (north_star: cell indices bit-exact)
            early = np.concatenate([np.load(tmp_path / f"early_{r}.npy") for r in range(world)])
            match_and_compare(early, ref.particles(), st.smoothing_radius, max_key_flips=0.0)
    # step 5: elementwise while ULP noise is small.  Whole lattice columns of this scene sit EXACTLY on a cell
    # boundary (x = k * 0.2), so from step 3 on a 1-ulp difference in x (summation order differs between a slab's
    # sort and the single-domain sort) puts such a particle in the neighbouring cell: measured 4 / 3 / 12 of 1024
    # at steps 3 / 4 / 5, 0 at steps 1-2
    match_and_compare(got, ref.particles(), st.smoothing_radius, max_key_flips=0.02)
    late = np.concatenate([np.load(tmp_path / f"late_{r}.npy") for r in range(world)])
    for _ in range(more):
        ref.step(tick)
    assert_statistics_close(late, ref.particles(), n)                     # then statistics (chaotic scene)


WORKER_FRONT = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import multi
from oracle import oracle as O
from tests.slab_oracle import OracleSlabEngine
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n, steps, speed, floor, mode = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
if mode == "undersized":
    multi.travel_margin = lambda *a, **k: 0          # the round-1 behaviour: a fixed margin, blind to the fluid's speed
st, off, tick = g.dam_break_2d(n)
tick.gravity = g.Vec2(0.0, 0.0)
hist, gw = multi.lattice_histogram(g, st, off)
bounds = multi.partition_columns(hist, world)
lat = g.reference_lattice(st, off)
lat["velocity"][:, 0] = speed                        # a front running towards the +x wall
margin0 = max(floor, multi.travel_margin(speed, 0.0, tick.delta, st.smoothing_radius, 4)) if mode != "undersized" else floor
bounds = multi.trim_outer_edges(bounds, hist, margin0)
tr = multi.Transport(rank, world, multi.HEADER_BYTES + multi.RECORD_BYTES * 8192)
eng = OracleSlabEngine(O, st, bounds, rank, world, tr)
cols = multi.global_columns(lat["position"][:, 0], st.size.x, st.smoothing_radius)
eng.upload_owned(lat[(cols >= bounds[rank]) & (cols < bounds[rank + 1])])
drv = multi.SlabDriver(eng, tr, bounds, gw, rebalance_every=4, max_shift=1, trim_margin=floor)
try:
    for _ in range(steps):
        drv.step(tick)
except multi.SlabProtocolError as e:
    print("PROTOCOL", e, flush=True)
    dist.destroy_process_group()
    sys.exit(7)
tot = torch.tensor([len(eng.owned_particles())])
dist.all_reduce(tot)
assert int(tot.item()) == n, (int(tot.item()), n)
assert drv.last_margin > floor, drv.last_margin       # the margin followed the measured speed
dist.barrier()
dist.destroy_process_group()
'''


def _run_front(tmp_path, mode, port):
    script = tmp_path / "worker_front.py"
    script.write_text(WORKER_FRONT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="1")
    # 4096 particles, 12 steps, 30 units/s = 1.25 columns per step = 5 per re-balancing interval, margin floor 2
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, "4096", "12", "30.0", "2", mode],
                              env=dict(env, RANK=str(r)), cwd=ROOT, stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    return [p.returncode for p in procs], outs


def test_gloo_fast_front_margin_follows_speed(fs, orc, tmp_path):
    """a front faster than a FIXED outer-edge margin used to be deleted silently.  The margin now comes
    from the all-reduced largest speed x the steps to the next re-trim: nothing is lost, counts are conserved."""
    rcs, outs = _run_front(tmp_path, "follow", 29521)
    assert rcs == [0, 0], outs


def test_gloo_undersized_margin_raises_protocol_error(fs, orc, tmp_path):
    """... and when particles ARE lost (margin forced back to the fixed floor) every rank raises at the next
    re-balancing step instead of running on with fewer particles."""
    rcs, outs = _run_front(tmp_path, "undersized", 29522)
    assert rcs == [7, 7], outs
    assert all("PROTOCOL" in o for o in outs)


def test_travel_margin_bounds():
    from gpu_fluid_simulation_amd import multi
    f = multi.travel_margin
    assert f(0.0, 0.0, 1 / 120, 0.2, 64) == 2                                  # nothing moves: just the halo tolerance
    assert f(127.0, 9.81, 1 / 120, 0.2, 64) >= int(np.ceil(64 * 127.0 / 120 / 0.2))   # covers the plain advection distance
    assert f(1e9, 0.0, 1 / 120, 0.2, 64) == int(np.ceil(64 * 500.0 / 120 / 0.2)) + 2  # never beyond the speed clamp
    assert f(30.0, 0.0, 1 / 120, 0.2, 4) < f(30.0, 0.0, 1 / 120, 0.2, 64)


# ---- properties of the pure partition logic (hypothesis) --------------------------------------------------------
from hypothesis import given, settings, strategies as hst


@settings(max_examples=200, deadline=None)
@given(hst.integers(2, 8), hst.lists(hst.integers(0, 50), min_size=40, max_size=300), hst.integers(0, 40),
       hst.integers(1, 5))
def test_partition_rebalance_trim_properties(world, counts, margin, max_shift):
    from gpu_fluid_simulation_amd import multi
    hist = np.array(counts, dtype=np.int64)
    gw = len(hist)
    b = multi.partition_columns(hist, world)
    assert len(b) == world + 1 and b[0] == 0 and b[-1] == gw
    assert all(b[k + 1] - b[k] >= 4 for k in range(world)), b          # every slab at least 4 columns wide
    # re-balancing never moves an interior boundary by more than max_shift and keeps the minimum width
    shifted = list(b)
    rng = np.random.default_rng(int(hist.sum()) + world)
    hist2 = np.roll(hist, int(rng.integers(-10, 10)))
    nb = multi.rebalance_boundaries(shifted, hist2, max_shift)
    assert nb[0] == b[0] and nb[-1] == b[-1]
    assert all(nb[k + 1] - nb[k] >= 4 for k in range(world)), nb
    assert all(abs(nb[k] - b[k]) <= max_shift + 4 for k in range(1, world))   # +4: the minimum-width repair may add to it
    # trimming only touches the two outer edges, stays inside the walls, never cuts into a neighbour,
    # and keeps every occupied column owned
    t = multi.trim_outer_edges(nb, hist2, margin)
    assert t[1:-1] == nb[1:-1]
    assert 0 <= t[0] <= nb[1] - 4 and nb[-2] + 4 <= t[-1] <= gw
    occ = np.nonzero(hist2)[0]
    if margin > 0 and occ.size:
        assert t[0] <= occ[0] and t[-1] >= occ[-1] + 1
        assert t[0] == max(0, min(occ[0] - margin, nb[1] - 4)) or t[0] == nb[1] - 4
    else:
        assert t[0] == 0 and t[-1] == gw
