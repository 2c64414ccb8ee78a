"""Pins the CPU oracle (oracle/sph_oracle.cpp).  The reference holds no tests or golden
vectors, so the pins are: known answers derived from the reference source
(SURVEY.md §A.7), an independent pure-Python f32 restatement (tests/pyref.py), structural
properties, and the committed fixture tests/golden/ (made by tests/golden/make_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import gpu_fluid_simulation_amd as g   # only for the ABI structs (ctypes layouts); no compute here
from tests import pyref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_struct_layouts():
    # ParticleInstance 32 B (simulation.rs:126-135), SimulationUniform 120 B (:53-90)
    assert g.PARTICLE_DTYPE.itemsize == 32
    assert [g.PARTICLE_DTYPE.fields[k][1] for k in ("position", "predicted_position", "velocity", "density", "grid")] \
        == [0, 8, 16, 24, 28]
    assert C.sizeof(g.Uniform) == 120
    offs = {n: getattr(g.Uniform, n).offset for n, _ in g.Uniform._fields_}
    assert offs["gravity"] == 16 and offs["bounds"] == 24 and offs["mouse_pos"] == 32
    assert offs["smoothing_radius"] == 40 and offs["poly6_kernel_volume"] == 72
    assert offs["spiky_kernel_derivative"] == 84 and offs["viscosity_kernel"] == 88
    assert offs["mouse_state"] == 92 and offs["grid_w"] == 104 and offs["texture_size"] == 112


def test_kernel_constants_known_answers(orc):
    # h = 0.2 -> 4/(pi h^8) = 497359.197, spiky = 2387.3241, visc = 298.41552
    st = g.SimulationSettings(4096, 0.1, 0.2, (53, 53))
    u = g.Uniform()
    t = g.default_tick_settings()
    orc.lib().orc_build_uniform(C.addressof(st), C.addressof(t), 7, C.addressof(u))
    assert u.frame_time == 7 and u.particle_count == 4096
    assert u.poly6_kernel_volume == pytest.approx(497359.197, rel=2e-6)
    assert u.spiky_kernel_derivative == pytest.approx(2387.3241, rel=1e-6)
    assert u.viscosity_kernel == pytest.approx(298.41552, rel=1e-6)
    assert u.sqr_radius == np.float32(0.2) * np.float32(0.2)
    assert (u.grid_w, u.grid_h) == (267, 267)          # 53.0/0.2 in f32 = 265.0 -> +2
    assert (u.texture_size.x, u.texture_size.y) == (1024.0, 1024.0)
    assert u.delta == np.float32(1.0) / np.float32(120.0)


def test_poly6_known_answers(orc):
    L = orc.lib()
    assert L.orc_poly6_value(0.2, 0.0) == pytest.approx(31.830989, rel=1e-6)
    assert L.orc_poly6_value(0.2, np.float32(0.1) ** 2) == pytest.approx(13.428698, rel=2e-6)
    assert L.orc_poly6_value(0.2, 2 * np.float32(0.1) ** 2) == pytest.approx(3.9788736, rel=3e-6)
    assert L.orc_poly6_value(0.2, np.float32(0.2) ** 2) == 0.0
    assert L.orc_poly6_value(0.2, 0.05) == 0.0          # r2 > h2 -> exactly +0


@pytest.mark.parametrize("n,size,grid", [
    (4096, (12.8, 8.0), (66, 42)), (1 << 20, (204.8, 128.0), (1026, 642)),
    (1 << 24, (819.2, 512.0), (4098, 2562)), (1 << 26, (1638.4, 1024.0), (8194, 5122)),
    (100_000, (53.0, 53.0), (267, 267)),
])
def test_grid_dims(orc, n, size, grid):
    st = g.SimulationSettings(n, 0.1, 0.2, size)
    w, h = C.c_uint32(), C.c_uint32()
    orc.lib().orc_grid_dims(C.addressof(st), C.addressof(w), C.addressof(h))
    assert (w.value, h.value) == grid


def _lattice(orc, n, off=(0.0, 0.0)):
    st = g.SimulationSettings(n, 0.1, 0.2, (53, 53))
    out = np.zeros(n, dtype=g.PARTICLE_DTYPE)
    orc.lib().orc_lattice(C.addressof(st), off[0], off[1], out.ctypes.data, n)
    return out


def test_lattice_known_extents(orc):
    # N = 4096 -> x in [-3.15, 3.15], y in [-3.19922, 3.10078]
    p = _lattice(orc, 4096)
    assert p["position"][:, 0].min() == pytest.approx(-3.15, abs=1e-6)
    assert p["position"][:, 0].max() == pytest.approx(3.15, abs=1e-6)
    assert p["position"][:, 1].min() == pytest.approx(-3.19922, abs=1e-5)
    assert p["position"][:, 1].max() == pytest.approx(3.10078, abs=1e-5)
    assert np.array_equal(p["position"], p["predicted_position"])
    assert not p["velocity"].any() and not p["density"].any() and not p["grid"].any()


def test_lattice_quirks_ragged_and_large(orc):
    # x uses truncated sqrt(N), y the un-truncated one
    p = _lattice(orc, 100_000)
    ppr = np.sqrt(np.float32(100_000))
    xs = np.unique(p["position"][:, 0])
    assert xs.shape[0] == 316                       # `as usize` truncation of 316.22775
    want_y = (np.floor(np.float32(99_999) / ppr) - ((np.float32(100_000) - 1) / ppr + 1) * np.float32(0.5)
              + np.float32(0.5)) * np.float32(0.1)
    assert p["position"][-1, 1] == want_y
    big = _lattice(orc, 1 << 24)
    assert big["position"][:, 0].min() == pytest.approx(-204.75, abs=1e-4)
    assert big["position"][:, 0].max() == pytest.approx(204.75, abs=1e-4)
    assert big["position"][:, 1].min() == pytest.approx(-204.8, abs=1e-4)
    assert big["position"][:, 1].max() == pytest.approx(204.7, abs=1e-4)


@pytest.mark.parametrize("n,count", [(4096, 78), (1 << 20, 210), (1 << 24, 300), (1 << 26, 351), (5000, 91), (2, 1)])
def test_sort_schedule_counts(orc, n, count):
    assert orc.lib().orc_sort_schedule(n, None, 0) == count
    arr = (g.SortStep * count)()
    orc.lib().orc_sort_schedule(n, arr, count)
    assert (arr[0].group_width, arr[0].group_height, arr[0].step_index, arr[0].num_values) == (1, 1, 0, n)
    if count >= 3:
        assert (arr[1].group_width, arr[1].group_height, arr[1].step_index) == (2, 3, 0)
        assert (arr[2].group_width, arr[2].group_height, arr[2].step_index) == (1, 1, 1)


@pytest.mark.parametrize("n", [2, 3, 5, 100, 127, 128, 129, 1000, 4096, 5000])
def test_bitonic_network_sorts_any_n(orc, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, max(2, n // 4), size=n, dtype=np.uint32)
    out, perm = orc.bitonic_keys(keys)
    assert np.all(out[:-1] <= out[1:])
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32))
    assert np.array_equal(keys[perm], out)


def test_bitonic_matches_wgsl_emulation_and_is_unstable(orc):
    # independent emulation of sort.wgsl:27-51 in Python; the network is NOT stable
    rng = np.random.default_rng(7)
    unstable = False
    for n in (5, 37, 128, 300):
        keys = rng.integers(0, 6, size=n, dtype=np.uint32)
        rec = [(int(k), i) for i, k in enumerate(keys)]
        pyref.bitonic(rec, lambda r: r[0], n)
        out, perm = orc.bitonic_keys(keys)
        assert [r[1] for r in rec] == perm.tolist()
        stable = np.argsort(keys, kind="stable")
        unstable |= not np.array_equal(stable, perm)
    assert unstable


def _scene(orc, n, jitter_seed=None, steps=0, **tick_over):
    st, off, tick = g.dam_break_2d(n)
    for k, v in tick_over.items():
        setattr(tick, k, v)
    o = orc.OracleSim(st, off)
    if jitter_seed is not None:
        rng = np.random.default_rng(jitter_seed)
        v = o.particles_view()
        j = rng.uniform(-0.025, 0.025, size=(n, 2)).astype(np.float32)
        v["position"] += j
        v["predicted_position"] = v["position"]
        v["velocity"] = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
    for _ in range(steps):
        o.step(tick)
    return o, st, off, tick


def test_interior_lattice_density_known_answer(orc):
    # 31.831 + 4*13.4287 + 4*3.97887 = 101.4609 for an interior lattice particle
    st = g.SimulationSettings(4096, 0.1, 0.2, (53, 53))
    o = orc.OracleSim(st)
    t = g.default_tick_settings()
    o.begin_tick(t); o.predict(); o.spatial_lookup(); o.sort(); o.cell_starts(); o.density(1)
    rho = o.particles()["density"]
    assert np.median(rho) == pytest.approx(101.4609, rel=2e-5)
    assert rho.max() == pytest.approx(101.4609, rel=2e-5)


def test_density_7x7_equals_3x3_bitwise(orc):
    # SURVEY A.4 — the reference sweeps 7x7 cells (funcs.wgsl:161-162); 3x3 is bit-identical
    for steps in (0, 30):
        o, st, off, tick = _scene(orc, 4096, jitter_seed=3, steps=steps)
        o.begin_tick(tick); o.predict(); o.spatial_lookup(); o.sort(); o.cell_starts()
        o.density(3)
        a = o.particles()["density"].copy()
        o.density(1)
        b = o.particles()["density"]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_stale_min_cell_quirk(orc):
    # sorted index 0 never writes its cell start, the table is never cleared.
    o, st, off, tick = _scene(orc, 4096)
    o.step(tick)
    p = o.particles()
    si = o.start_indices()
    cmin = p["grid"][0]
    assert si[cmin] == 0                       # never written, still the initial zero
    occupied, first = np.unique(p["grid"], return_index=True)
    assert np.array_equal(si[occupied[1:]], first[1:].astype(np.uint32))
    # poison the min cell's entry: its particles become invisible as neighbours
    v = o.start_indices_view()
    v[cmin] = 3
    before = o.particles()["density"].copy()
    o.begin_tick(tick); o.predict(); o.spatial_lookup(); o.sort(); o.cell_starts()
    assert o.start_indices()[o.particles()["grid"][0]] == 3   # still stale after the pass
    del before


def test_oracle_matches_python_restatement(orc):
    # tiny N, 2 steps, bit-for-bit against tests/pyref.py (7x7 sweep, full network)
    f = np.float32
    n = 100
    st = g.SimulationSettings(n, 0.1, 0.2, (4.0, 3.0))
    tick = g.default_tick_settings(gravity=(0.5, 9.81))
    o = orc.OracleSim(st)
    rng = np.random.default_rng(11)
    v = o.particles_view()
    v["position"] += rng.uniform(-0.03, 0.03, size=(n, 2)).astype(f)
    v["predicted_position"] = v["position"]
    v["velocity"] = rng.uniform(-2, 2, size=(n, 2)).astype(f)
    parts = [dict(pos=(f(q["position"][0]), f(q["position"][1])), pred=(f(0), f(0)),
                  vel=(f(q["velocity"][0]), f(q["velocity"][1])), density=f(0), grid=0) for q in v]
    gw, gh = o.grid_dims
    si = np.zeros(gw * gh, dtype=np.uint32)
    for s in range(2):
        o.step(tick)
        u = g.Uniform.from_buffer_copy(o.uniform_bytes())
        ud = dict(bounds=(f(u.bounds.x), f(u.bounds.y)), h=f(u.smoothing_radius), dt=f(u.delta), grid_w=u.grid_w,
                  mass=f(u.particle_mass), pow_h8=f(np.float32(u.smoothing_radius) ** np.float32(8.0)),
                  k=f(u.pressure_constant), rho0=f(u.rest_density), frame=u.frame_time, sqr_radius=f(u.sqr_radius),
                  spiky=f(u.spiky_kernel_derivative), visc=f(u.viscosity_kernel),
                  visc_coeff=f(u.viscosity_coefficient), gravity=(f(u.gravity.x), f(u.gravity.y)),
                  damping=f(u.damping_factor))
        pyref.step(parts, si, ud)
        got = o.particles()
        assert [p["grid"] for p in parts] == got["grid"].tolist()
        for i, p in enumerate(parts):
            assert (p["pos"][0], p["pos"][1]) == tuple(got["position"][i]), (s, i)
            assert (p["vel"][0], p["vel"][1]) == tuple(got["velocity"][i]), (s, i)
            assert p["density"] == got["density"][i]
        assert np.array_equal(si, o.start_indices())


def _pyref_uniform(u, f, **extra):
    d = dict(bounds=(f(u.bounds.x), f(u.bounds.y)), h=f(u.smoothing_radius), dt=f(u.delta), grid_w=u.grid_w,
             mass=f(u.particle_mass), pow_h8=f(np.float32(u.smoothing_radius) ** np.float32(8.0)),
             k=f(u.pressure_constant), rho0=f(u.rest_density), frame=u.frame_time, sqr_radius=f(u.sqr_radius),
             spiky=f(u.spiky_kernel_derivative), visc=f(u.viscosity_kernel),
             visc_coeff=f(u.viscosity_coefficient), gravity=(f(u.gravity.x), f(u.gravity.y)),
             damping=f(u.damping_factor), mouse_state=int(u.mouse_state), mouse_pos=(f(u.mouse_pos.x), f(u.mouse_pos.y)),
             mouse_radius=f(u.mouse_force_radius), mouse_power=f(u.mouse_force_power),
             texture_size=(f(u.texture_size.x), f(u.texture_size.y)))
    d.update(extra)
    return d


def _run_both(orc, o, parts, tick, steps, texture=None):
    """Step the C++ oracle and the pure-Python restatement side by side; everything must agree bit for bit."""
    f = np.float32
    gw, gh = o.grid_dims
    si = np.zeros(gw * gh, dtype=np.uint32)
    for s in range(steps):
        o.step(tick)
        u = g.Uniform.from_buffer_copy(o.uniform_bytes())
        pyref.step(parts, si, _pyref_uniform(u, f, texture=texture))
        got = o.particles()
        assert [p["grid"] for p in parts] == got["grid"].tolist(), s
        for i, p in enumerate(parts):
            a = np.array([p["pos"][0], p["pos"][1], p["vel"][0], p["vel"][1], p["density"]], dtype=f).view(np.uint32)
            b = np.array([got["position"][i][0], got["position"][i][1], got["velocity"][i][0], got["velocity"][i][1],
                          got["density"][i]], dtype=f).view(np.uint32)
            assert np.array_equal(a, b), (s, i, a, b)
        assert np.array_equal(si, o.start_indices())
    return o.particles()


def _parts_of(view):
    f = np.float32
    return [dict(pos=(f(q["position"][0]), f(q["position"][1])), pred=(f(0), f(0)),
                 vel=(f(q["velocity"][0]), f(q["velocity"][1])), density=f(0), grid=0) for q in view]


def test_oracle_matches_python_restatement_mouse_and_walls(orc):
    """compute.wgsl:99-108 (mouse impulse, both buttons) and :143-153 (wall clamp with damped reflection of a MOVING
    particle): 150 particles thrown at the walls while the mouse pulls / pushes; 3 steps, bit for bit."""
    f = np.float32
    n = 150
    st = g.SimulationSettings(n, 0.1, 0.2, (2.4, 2.0))
    for state in (1, -1):
        tick = g.default_tick_settings(gravity=(0.0, 9.81), mouse_state=state, mouse_pos=(0.3, -0.2),
                                       mouse_force_radius=0.9, mouse_force_power=40.0)
        o = orc.OracleSim(st)
        rng = np.random.default_rng(5 + state)
        v = o.particles_view()
        v["position"] += rng.uniform(-0.03, 0.03, size=(n, 2)).astype(f)
        v["predicted_position"] = v["position"]
        v["velocity"] = rng.uniform(-40, 40, size=(n, 2)).astype(f)       # 0.33 units / step: walls within reach
        out = _run_both(orc, o, _parts_of(v), tick, 3)
        bs = np.array([1.2, 1.0], dtype=f)
        assert (np.abs(out["position"]) == bs).any(), "no particle reached a wall: the bounce branch did not run"


def test_oracle_matches_python_restatement_obstacle_field(orc):
    """compute.wgsl:127-140: texture lookup at the predicted position, push-out by the field vector, damped removal
    of the normal velocity — with a non-zero 16x12 field (zero in one corner, so both branches run)."""
    f = np.float32
    n = 120
    st = g.SimulationSettings(n, 0.1, 0.2, (3.0, 2.4), (16, 12))
    tick = g.default_tick_settings(gravity=(0.3, 9.81))
    o = orc.OracleSim(st)
    rng = np.random.default_rng(31)
    v = o.particles_view()
    v["position"] += rng.uniform(-0.03, 0.03, size=(n, 2)).astype(f)
    v["predicted_position"] = v["position"]
    v["velocity"] = rng.uniform(-3, 3, size=(n, 2)).astype(f)
    tex = rng.uniform(-0.02, 0.02, size=(12, 16, 2)).astype(f)
    tex[:6, :8] = 0.0
    o.texture_view()[:] = tex
    before = o.particles()["position"].copy()
    out = _run_both(orc, o, _parts_of(v), tick, 3, texture=[(a, b) for a, b in tex.reshape(-1, 2)])
    assert not np.array_equal(before, out["position"])
    nz = (tex.reshape(-1, 2) != 0).any(axis=1).mean()
    assert 0.2 < nz < 0.9


def test_oracle_matches_python_restatement_prng_nan_and_clamp(orc):
    """Coincident particles (r == 0 -> xorshift32 direction, compute.wgsl:211-212 + funcs.wgsl:129-149, and the
    r == 0 viscosity constant funcs.wgsl:116), a NaN velocity (reset to 0, compute.wgsl:113-116) and a particle far
    above the 500 speed clamp (:118-122); 2 steps, bit for bit."""
    f = np.float32
    n = 64
    st = g.SimulationSettings(n, 0.1, 0.2, (3.0, 3.0))
    tick = g.default_tick_settings(gravity=(0.0, 9.81))
    o = orc.OracleSim(st)
    v = o.particles_view()
    v["position"][5] = v["position"][4]              # two coincident pairs
    v["position"][41] = v["position"][40]
    v["position"][42] = v["position"][40]            # ... and a coincident triple
    v["predicted_position"] = v["position"]
    v["velocity"][:] = 0
    v["velocity"][10] = (np.nan, 1.0)
    v["velocity"][20] = (9000.0, -7000.0)
    out = _run_both(orc, o, _parts_of(v), tick, 2)
    assert np.isfinite(out["velocity"]).all() and np.isfinite(out["position"]).all()
    assert np.sqrt((out["velocity"].astype(np.float64) ** 2).sum(axis=1)).max() <= 500.001


def test_stated_tolerance_against_f64_run(orc):
    """SURVEY.md §8c states the float tolerance for consumers that do not rely on bit equality: one step from an
    identical state with an identical permutation — density rel <= 1e-5, velocity / position abs <= 1e-4*h + rel
    1e-5.  Confirm it: the f32 oracle against the same algorithm evaluated in float64 (tests/pyref.py)."""
    f32, f64 = np.float32, np.float64
    n = 196
    st = g.SimulationSettings(n, 0.1, 0.2, (5.0, 4.0))
    tick = g.default_tick_settings(gravity=(0.5, 9.81))
    o = orc.OracleSim(st)
    rng = np.random.default_rng(23)
    v = o.particles_view()
    v["position"] += rng.uniform(-0.03, 0.03, size=(n, 2)).astype(f32)
    v["predicted_position"] = v["position"]
    v["velocity"] = rng.uniform(-2, 2, size=(n, 2)).astype(f32)
    parts = [dict(pos=(f64(q["position"][0]), f64(q["position"][1])), pred=(f64(0), f64(0)),
                  vel=(f64(q["velocity"][0]), f64(q["velocity"][1])), density=f64(0), grid=0) for q in v]
    gw, gh = o.grid_dims
    si = np.zeros(gw * gh, dtype=np.uint32)
    o.step(tick)
    u = g.Uniform.from_buffer_copy(o.uniform_bytes())
    h = f64(u.smoothing_radius)
    ud = dict(bounds=(f64(u.bounds.x), f64(u.bounds.y)), h=h, dt=f64(u.delta), grid_w=u.grid_w,
              mass=f64(u.particle_mass), pow_h8=h ** 8, k=f64(u.pressure_constant), rho0=f64(u.rest_density),
              frame=u.frame_time, sqr_radius=h * h, spiky=f64(u.spiky_kernel_derivative), visc=f64(u.viscosity_kernel),
              visc_coeff=f64(u.viscosity_coefficient), gravity=(f64(u.gravity.x), f64(u.gravity.y)),
              damping=f64(u.damping_factor))
    prev = pyref.set_float(f64)
    try:
        pyref.step(parts, si, ud)
    finally:
        pyref.set_float(prev)
    got = o.particles()
    assert [p["grid"] for p in parts] == got["grid"].tolist(), "the f64 run must see the same cells and permutation"
    rho64 = np.array([p["density"] for p in parts])
    pos64 = np.array([p["pos"] for p in parts])
    vel64 = np.array([p["vel"] for p in parts])
    hh = float(h)
    assert np.all(np.abs(got["density"] - rho64) <= 1e-5 * np.abs(rho64))
    assert np.all(np.abs(got["velocity"] - vel64) <= 1e-4 * hh + 1e-5 * np.abs(vel64))
    assert np.all(np.abs(got["position"] - pos64) <= 1e-4 * hh + 1e-5 * np.abs(pos64))
    # and the margin actually observed (documented in DESIGN.md §2)
    print("max rel density err", np.max(np.abs(got["density"] - rho64) / np.abs(rho64)),
          "max abs vel err", np.max(np.abs(got["velocity"] - vel64)), "max abs pos err", np.max(np.abs(got["position"] - pos64)))


def test_invariants_long_run(orc):
    o, st, off, tick = _scene(orc, 4096, steps=150)
    p = o.particles()
    assert p.shape[0] == 4096
    assert np.isfinite(p["position"]).all() and np.isfinite(p["velocity"]).all()
    assert np.abs(p["position"][:, 0]).max() <= st.size.x / 2 and np.abs(p["position"][:, 1]).max() <= st.size.y / 2
    assert np.all(p["grid"][:-1] <= p["grid"][1:])
    assert p["density"].min() >= np.float32(0.1)
    assert np.hypot(*p["velocity"].T).max() <= 500.0


def test_coincident_particles_take_prng_path(orc):
    # dst == 0 -> random direction from xorshift32 (compute.wgsl:211-212); must stay finite
    o, st, off, tick = _scene(orc, 4096)
    v = o.particles_view()
    v["position"][1] = v["position"][0]
    v["predicted_position"][1] = v["position"][0]
    o.step(tick)
    p = o.particles()
    assert np.isfinite(p["velocity"]).all()


def test_nan_reset_and_speed_clamp(orc):
    o, st, off, tick = _scene(orc, 4096)
    v = o.particles_view()
    v["velocity"][5] = (np.nan, 1.0)
    v["velocity"][9] = (9000.0, 0.0)
    o.step(tick)
    p = o.particles()
    assert np.isfinite(p["velocity"]).all()
    assert np.hypot(*p["velocity"].T).max() <= 500.0 * (1 + 1e-6)


def test_n_le_1_rejected(orc):
    for n in (0, 1):
        st = g.SimulationSettings(n, 0.1, 0.2, (53, 53))
        with pytest.raises(ValueError):
            orc.OracleSim(st)


def test_golden_fixture(orc):
    path = os.path.join(GOLD, "dam_break_4096.npz")
    z = np.load(path)   # allow_pickle=False default
    o, st, off, tick = _scene(orc, 4096)
    for s in range(int(z["steps"])):
        o.step(tick)
        assert np.array_equal(o.particles().view(np.uint8), z[f"particles_{s}"].view(np.uint8)), s
        assert np.array_equal(o.start_indices(), z[f"start_indices_{s}"]), s
