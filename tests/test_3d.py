"""3D extension (27-cell path).  The reference is 2D only: the oracle here is this repo's own
statement (oracle/sph_oracle3d.cpp) — parity with a reference is undefined;
what is pinned: analytic kernel constants / interior lattice density, and HIP == oracle bit for bit."""
import os

import numpy as np
import pytest

FLOATS = ("position", "predicted_position", "velocity", "density")


def test_oracle3d_known_answers(fs, orc):
    # h = 0.2: 315/(64 pi h^9) = 3.05992e6, 15/(pi h^5) = 14920.8, 15/(2 pi h^3) = 298.4155
    st = fs.Settings3(12 ** 3, 0.1, 0.2, fs.Vec3(20.0, 20.0, 20.0))
    tick = fs.TickSettings3(float(np.float32(1) / np.float32(120)), fs.Vec3(0, 0, 0), 1.0, 50.0, 0.0, 0.1, 25.0)
    o = orc.OracleSim3D(st)
    assert o.grid_dims == (102, 102, 102)
    o.step(tick)
    poly6, spiky, visc = o.constants()
    assert poly6 == pytest.approx(3.0599245e6, rel=2e-6)
    assert spiky == pytest.approx(14920.775, rel=2e-6)
    assert visc == pytest.approx(298.41552, rel=2e-6)
    # interior lattice particle, s = 0.1: W(0) + 6 W(0.1) + 12 W(0.1 sqrt2) + 8 W(0.1 sqrt3) + 6 W(0.2)=0
    h2 = 0.04
    want = poly6 * (h2 ** 3 + 6 * (h2 - 0.01) ** 3 + 12 * (h2 - 0.02) ** 3 + 8 * (h2 - 0.03) ** 3)
    assert o.particles()["density"].max() == pytest.approx(want, rel=1e-4)
    p = o.particles()
    assert np.all(p["grid"][:-1] <= p["grid"][1:])


def test_oracle3d_lattice_and_invariants(fs, orc):
    st, off, tick = fs.dam_break_3d(10 ** 3)
    o = orc.OracleSim3D(st, off)
    p0 = o.particles()
    assert np.unique(p0["position"], axis=0).shape[0] == 1000
    assert p0["position"][:, 0].min() == pytest.approx(-st.size.x / 2 + 0.1, abs=1e-5)   # one spacing off the wall
    for _ in range(60):
        o.step(tick)
    p = o.particles()
    assert np.isfinite(p["position"]).all() and np.isfinite(p["velocity"]).all()
    for a, b in enumerate((st.size.x, st.size.y, st.size.z)):
        assert np.abs(p["position"][:, a]).max() <= b / 2
    assert p["density"].min() >= np.float32(0.1)


def test_lattice3d_matches_oracle(fs, orc):
    import ctypes as C
    st, off, tick = fs.dam_break_3d(9 ** 3)
    got = fs.reference_lattice_3d(st, off)
    want = np.zeros(9 ** 3, dtype=fs.PARTICLE3_DTYPE)
    orc.lib().orc3_lattice(C.addressof(st), off[0], off[1], off[2], want.ctypes.data, want.shape[0])
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def _assert_equal3(got, want, ctx):
    assert np.array_equal(got["grid"], want["grid"]), f"{ctx}: cell keys differ"
    for f in FLOATS:
        a, b = got[f].view(np.uint32), want[f].view(np.uint32)
        if not np.array_equal(a, b):
            err = np.abs(got[f].astype(np.float64) - want[f].astype(np.float64)).max()
            raise AssertionError(f"{ctx}: {f} not bit-exact ({int((a != b).sum())} words), max abs err {err:g}")


@pytest.mark.gpu
@pytest.mark.parametrize("side,seed,steps", [(16, None, 10), (20, 5, 5), (3, 1, 4), (33, 9, 2)])
def test_3d_parity_with_oracle(fs, orc, side, seed, steps):
    n = side ** 3
    st, off, tick = fs.dam_break_3d(n)
    sim = fs.FluidSimulation3D(st, device=0, initial_offset=off)
    ref = orc.OracleSim3D(st, off)
    if seed is not None:
        rng = np.random.default_rng(seed)
        p = ref.particles()
        p["position"] += rng.uniform(-0.025, 0.025, size=(n, 3)).astype(np.float32)
        p["predicted_position"] = p["position"]
        p["velocity"] = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
        ref.set_particles(p)
        sim.upload_particles(p)
    assert sim.grid_dims == ref.grid_dims
    _assert_equal3(sim.download_particles(), ref.particles(), "initial")
    for s in range(steps):
        sim.tick(tick)
        ref.step(tick)
        _assert_equal3(sim.download_particles(), ref.particles(), f"3d side {side} step {s}")
    assert sim.tick_count == steps


@pytest.mark.gpu
def test_3d_coincident_and_guards(fs, orc):
    n = 12 ** 3
    st, off, tick = fs.dam_break_3d(n)
    sim = fs.FluidSimulation3D(st, device=0, initial_offset=off)
    ref = orc.OracleSim3D(st, off)
    p = ref.particles()
    p["position"][1:4] = p["position"][0]
    p["predicted_position"][1:4] = p["position"][0]
    p["velocity"][7] = (np.nan, 0, 1)
    p["velocity"][9] = (9000, -9000, 100)
    ref.set_particles(p); sim.upload_particles(p)
    for s in range(3):
        sim.tick(tick); ref.step(tick)
        _assert_equal3(sim.download_particles(), ref.particles(), f"guards step {s}")


@pytest.mark.gpu
def test_3d_8m_properties(fs):
    """BASELINE configs[3] size: sortedness, finite state, analytic interior density."""
    n = 200 ** 3
    st, off, tick = fs.dam_break_3d(n)
    sim = fs.FluidSimulation3D(st, device=0, initial_offset=off)
    assert sim.grid_dims == (202, 127, 103)
    for _ in range(3):
        sim.tick(tick)
    p = sim.download_particles()
    assert np.all(p["grid"][:-1] <= p["grid"][1:])
    assert np.isfinite(p["position"]).all() and np.isfinite(p["velocity"]).all()
    assert np.median(p["density"]) == pytest.approx(1009.8, rel=2e-3)


@pytest.mark.gpu
def test_3d_dense_cluster(fs, orc):
    """Hot cells in 3D: list flushes and the global fallback of the staged density path."""
    n = 12 ** 3
    st, off, tick = fs.dam_break_3d(n)
    sim = fs.FluidSimulation3D(st, device=0, initial_offset=off)
    ref = orc.OracleSim3D(st, off)
    rng = np.random.default_rng(23)
    p = ref.particles()
    idx = rng.choice(n, 900, replace=False)
    p["position"][idx] = rng.uniform(-0.25, 0.25, size=(900, 3)).astype(np.float32) + np.float32([0.3, 0.2, 0.1])
    p["predicted_position"] = p["position"]
    ref.set_particles(p); sim.upload_particles(p)
    for s in range(2):
        sim.tick(tick); ref.step(tick)
        _assert_equal3(sim.download_particles(), ref.particles(), f"3d cluster step {s}")


def test_oracle3d_is_thread_count_independent(fs, orc):
    """The 3D oracle's OpenMP loops write one record per iteration: 1 thread == all threads, bit for bit."""
    st, off, tick = fs.dam_break_3d(14 ** 3)
    outs = []
    for threads in (1, max(2, min(8, orc.max_threads()))):
        orc.set_threads(threads)
        o = orc.OracleSim3D(st, off)
        rng = np.random.default_rng(4)
        p = o.particles()
        p["position"] += rng.uniform(-0.03, 0.03, size=p["position"].shape).astype(np.float32)
        p["predicted_position"] = p["position"]
        o.set_particles(p)
        for _ in range(3):
            o.step(tick)
        outs.append(o.particles().copy())
        orc.set_threads(1)
    assert np.array_equal(outs[0].view(np.uint8), outs[1].view(np.uint8))


@pytest.mark.gpu
def test_3d_8m_full_state_matches_oracle(fs, orc):
    """BASELINE configs[3] at its full size: one step of the 200^3 dam break on the GPU, EVERY field of EVERY particle
    bit for bit against the 3D oracle run on all host cores (keys, positions, predicted positions, velocities,
    densities).  compute.wgsl:45-157 shapes; the 3D statement itself is build-defined (no reference counterpart)."""
    import bench
    n = 200 ** 3
    st, off, tick = fs.dam_break_3d(n)
    sim = fs.FluidSimulation3D(st, device=0, initial_offset=off)
    sim.tick(tick)
    got = sim.download_particles()
    sim.close()
    orc.set_threads(min(bench.usable_cores(), orc.max_threads()))
    try:
        ref = orc.OracleSim3D(st, off)
        ref.step(tick)
        _assert_equal3(got, ref.particles_view(), "3d 8M step 1")
    finally:
        orc.set_threads(1)


@pytest.mark.gpu
def test_3d_shuffled_upload_takes_the_wide_tile_path(fs, orc):
    """The same hand-over in 3D: the 8M scene's grid has 402 x 252 x 204 = 20.7M cells, so a permuted upload makes
    every 4096-element tile of the first sort kernel wide; the step after it must still equal the 3D oracle bit for bit."""
    import bench
    n = 200 ** 3
    st, off, tick = fs.dam_break_3d(n)
    sim = fs.FluidSimulation3D(st, device=0, initial_offset=off)
    p = sim.download_particles()
    p = p[np.random.default_rng(6).permutation(n)]
    sim.upload_particles(p)
    sim.tick(tick)
    got = sim.download_particles()
    sim.close()
    orc.set_threads(min(bench.usable_cores(), orc.max_threads()))
    try:
        ref = orc.OracleSim3D(st, off)
        ref.set_particles(p)
        ref.step(tick)
        _assert_equal3(got, ref.particles_view(), "3d 8M step 1 after a shuffled upload")
    finally:
        orc.set_threads(1)


@pytest.mark.gpu
@pytest.mark.parametrize("side,seed,pre", [(16, None, 0), (24, 7, 3), (33, 9, 10), (12, 2, 0)])
def test_3d_tolerance_mode_within_tolerance(fs, orc, side, seed, pre):
    """fs3_create_ex(FS_MATH_TOLERANCE): one step from an identical state against the 3D oracle — cell keys bit-exact
    (sort and reorder are untouched), predicted positions bit-exact, density rtol 1e-5, velocity rtol 1e-5 + atol 2e-5,
    position atol 1e-4 * h; `pre` strict steps first so the state is not a lattice."""
    n = side ** 3
    st, off, tick = fs.dam_break_3d(n)
    ref = orc.OracleSim3D(st, off)
    p = ref.particles()
    if seed is not None:
        rng = np.random.default_rng(seed)
        p["position"] += rng.uniform(-0.03, 0.03, size=(n, 3)).astype(np.float32)
        p["predicted_position"] = p["position"]
        p["velocity"] = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
        if side == 12:                                     # coincident particles: the PRNG direction is kept
            p["position"][1:4] = p["position"][0]
            p["predicted_position"][1:4] = p["position"][0]
        ref.set_particles(p)
    for _ in range(pre):
        ref.step(tick)
    state = ref.particles()
    sim = fs.FluidSimulation3D(st, device=0, initial_offset=off, math_mode=fs.FS_MATH_TOLERANCE)
    sim.upload_particles(state)
    sim.tick(tick)
    ref.step(tick)
    got, want = sim.download_particles(), ref.particles()
    assert np.array_equal(got["grid"], want["grid"]), "cell keys must stay bit-exact in tolerance mode"
    assert np.array_equal(got["predicted_position"].view(np.uint32), want["predicted_position"].view(np.uint32))
    h = float(st.smoothing_radius)
    np.testing.assert_allclose(got["density"], want["density"], rtol=1e-5)
    np.testing.assert_allclose(got["velocity"], want["velocity"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(got["position"], want["position"], rtol=0, atol=1e-4 * h)
    sim.close()


@pytest.mark.gpu
def test_3d_tolerance_mode_many_steps_stays_close(fs, orc):
    """20 tolerance-mode steps of a 27k-particle dam break: keys identical to the oracle while no particle sits within
    rounding distance of a cell face; rounding-level differences grow ~2.4x per step at the free surface, so
    after 20 steps only bulk statistics are compared (1 %)."""
    st, off, tick = fs.dam_break_3d(30 ** 3)
    sim = fs.FluidSimulation3D(st, device=0, initial_offset=off, math_mode=fs.FS_MATH_TOLERANCE)
    ref = orc.OracleSim3D(st, off)
    for _ in range(20):
        sim.tick(tick); ref.step(tick)
    got, want = sim.download_particles(), ref.particles()
    assert np.isfinite(got["position"]).all()
    assert abs(float(got["density"].mean()) / float(want["density"].mean()) - 1.0) < 1e-2
    assert abs(float(np.abs(got["velocity"]).mean()) / float(np.abs(want["velocity"]).mean()) - 1.0) < 1e-2
    assert np.abs(got["position"]).max() <= max(st.size.x, st.size.y, st.size.z) / 2
    sim.close()


@pytest.mark.gpu
def test_3d_mask_handoff_does_not_change_a_bit(fs, tmp_path):
    """k3_density hands its nine pass masks to k3_force (default); FS3_HANDOFF=0 makes the force pass scan the 216
    candidates itself as in round 2, FS3_SEPARATE_KEYGEN=1 brings back the separate predict+key launch, FS3_XCD_CHUNK_LOG2
    changes the workgroup -> block mapping.  Same bits."""
    import subprocess, sys, textwrap
    prog = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        import gpu_fluid_simulation_amd as fs
        st, off, tick = fs.dam_break_3d(40 ** 3)
        sim = fs.FluidSimulation3D(st, device=0, initial_offset=off)
        rng = np.random.default_rng(5)
        p = sim.download_particles()
        p["position"] += rng.uniform(-0.03, 0.03, size=p["position"].shape).astype(np.float32)
        p["predicted_position"] = p["position"]
        idx = rng.choice(p.shape[0], 1500, replace=False)       # a dense cluster: rows > 64 take the chunked sweep
        p["position"][idx] = rng.uniform(-0.2, 0.2, size=(1500, 3)).astype(np.float32)
        p["predicted_position"] = p["position"]
        sim.upload_particles(p)
        for _ in range(6):
            sim.tick(tick)
        np.save(sys.argv[1], sim.download_particles())
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    # FS3_XCD_CHUNK_LOG2: which workgroup takes which block of particles (XCD-aware mapping) — a permutation of the work
    for k, env in enumerate(({}, {"FS3_HANDOFF": "0"}, {"FS3_SEPARATE_KEYGEN": "1"}, {"FS3_XCD_CHUNK_LOG2": "0"},
                             {"FS3_XCD_CHUNK_LOG2": "3", "FS3_HANDOFF": "0"})):
        out = tmp_path / f"o{k}.npy"
        r = subprocess.run([sys.executable, "-c", prog, str(out)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    assert np.array_equal(outs[0].view(np.uint8), outs[1].view(np.uint8))
    assert np.array_equal(outs[0].view(np.uint8), outs[2].view(np.uint8))
    assert np.array_equal(outs[0].view(np.uint8), outs[3].view(np.uint8))
    assert np.array_equal(outs[0].view(np.uint8), outs[4].view(np.uint8))


@pytest.mark.parametrize("seed,coincident", [(1, False), (2, True)])
def test_oracle3d_equals_independent_python_restatement(fs, orc, seed, coincident):
    """The 3D statement has no reference counterpart, so the C++ 3D oracle is cross-checked by a second restatement in
    pure Python with np.float32 scalars (tests/pyref3d.py): 3 steps of ~90 jittered particles with velocities, optionally
    with a coincident triple (PRNG direction, r = 0 viscosity), bit for bit on every field."""
    import pyref3d
    f = np.float32
    side = 4 if coincident else 5
    n = side ** 3
    st = fs.Settings3(n, 0.1, 0.2, fs.Vec3(1.6, 1.2, 1.4))
    tick = fs.TickSettings3(float(f(1) / f(120)), fs.Vec3(0.3, 9.81, -0.2), 1.0, 50.0, 0.0, 0.1, 25.0)
    o = orc.OracleSim3D(st, (0.05, 0.1, -0.05))
    rng = np.random.default_rng(seed)
    p = o.particles()
    p["position"] += rng.uniform(-0.04, 0.04, size=(n, 3)).astype(f)
    if coincident:
        p["position"][1:3] = p["position"][0]
    p["predicted_position"] = p["position"]
    p["velocity"] = rng.uniform(-2, 2, size=(n, 3)).astype(f)
    if coincident:
        p["velocity"][5] = (400.0, -450.0, 300.0)            # the 500 clamp and a wall bounce
    o.set_particles(p)
    parts = [dict(pos=tuple(f(x) for x in r["position"]), pred=tuple(f(x) for x in r["predicted_position"]),
                  vel=tuple(f(x) for x in r["velocity"]), density=f(r["density"]), grid=int(r["grid"])) for r in p]
    for s in range(3):
        o.step(tick)
        poly6, spiky, visc = o.constants()
        u = dict(grid=o.grid_dims, h=f(st.smoothing_radius), dt=f(tick.delta), size=(f(st.size.x), f(st.size.y), f(st.size.z)),
                 mass=f(tick.mass), k=f(tick.pressure_constant), rho0=f(tick.rest_density), damping=f(tick.damping_factor),
                 visc_coeff=f(tick.viscosity_coefficient), gravity=(f(tick.gravity.x), f(tick.gravity.y), f(tick.gravity.z)),
                 poly6=f(poly6), spiky=f(spiky), visc=f(visc), tick=s + 1)
        pyref3d.step3(parts, u)
        want = o.particles()
        assert [q["grid"] for q in parts] == list(want["grid"]), f"step {s}: keys / order"
        for name, key in (("position", "pos"), ("predicted_position", "pred"), ("velocity", "vel")):
            got = np.array([q[key] for q in parts], dtype=f)
            assert np.array_equal(got.view(np.uint32), want[name].view(np.uint32)), f"step {s}: {name}"
        got = np.array([q["density"] for q in parts], dtype=f)
        assert np.array_equal(got.view(np.uint32), want["density"].view(np.uint32)), f"step {s}: density"
