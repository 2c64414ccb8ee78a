"""One-off fuzz: seeded random scenes (settings, tick constants, particle states incl. dense and coincident ones)
bit-compared with the oracle in both sort modes.  python tools/fuzz_parity.py [first_case] [cases]
The committed suite runs 16 such cases (tests/test_parity_gpu.py::test_random_configurations); this is the long form."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import gpu_fluid_simulation_amd as fs
from oracle import oracle as orc
from test_parity_gpu import assert_particles_equal

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t0 = time.time()
for case in range(first, first + cases):
    rng = np.random.default_rng(5000 + case)
    n = int(rng.integers(2, 30000)) if case % 7 else int(rng.integers(30000, 120000))
    h = float(rng.choice([0.05, 0.1, 0.2, 0.33, 0.5, 1.0]))
    # every fifth case is compressive: many particles per cell -> long sweep rows, chunked / unstaged paths
    spacing = float(h * (rng.uniform(0.04, 0.15) if case % 5 == 0 else rng.uniform(0.3, 0.9)))
    side = np.sqrt(n) * spacing
    size = (float(side * rng.uniform(1.2, 3.0) + 4 * h), float(side * rng.uniform(1.2, 3.0) + 4 * h))
    tex = (int(rng.choice([64, 256, 1024])), int(rng.choice([64, 128, 1024])))
    st = fs.SimulationSettings(n, spacing, h, size, tex)
    tick = fs.default_tick_settings(
        delta=float(rng.choice([1 / 240, 1 / 120, 1 / 60])), gravity=(float(rng.uniform(-5, 5)), float(rng.uniform(-10, 10))),
        mass=float(rng.uniform(0.5, 2.0)), pressure_constant=float(rng.choice([0.0, 5.0, 50.0, 500.0, 1e6])),
        rest_density=float(rng.choice([0.0, 1.0, 20.0, 1000.0])), damping_factor=float(rng.uniform(0.0, 0.9)),
        viscosity_coefficient=float(rng.choice([0.0, 5.0, 25.0, 1e4])), mouse_state=int(rng.choice([0, 0, 1, -1])),
        mouse_pos=(float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))), mouse_force_radius=float(rng.uniform(0.5, 5)))
    off = (float(rng.uniform(-0.2, 0.2) * size[0]), float(rng.uniform(-0.2, 0.2) * size[1]))
    steps = 3 if n > 30000 else 6
    for mode, stable in ((fs.FS_SORT_BITONIC, False), (fs.FS_SORT_COUNTING, True)):
        sim = fs.FluidSimulation(st, device=0, initial_offset=off, sort_mode=mode)
        ref = orc.OracleSim(st, off)
        p = ref.particles()
        p["position"] += rng.uniform(-0.3, 0.3, size=(n, 2)).astype(np.float32) * np.float32(spacing)
        if case % 4 == 1 and n > 20:                       # coincident groups -> PRNG path
            k = int(rng.integers(2, 8))
            p["position"][1:k] = p["position"][0]
        p["predicted_position"] = p["position"]
        p["velocity"] = (rng.standard_normal((n, 2)) * float(rng.choice([0.0, 1e-6, 2.0, 300.0]))).astype(np.float32)
        ref.set_particles(p); sim.upload_particles(p)
        if case % 3 == 0:
            field = np.zeros((tex[1], tex[0], 2), dtype=np.float32)
            field[tex[1] // 3: tex[1] // 2, tex[0] // 4: tex[0] // 2] = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)))
            sim.upload_force_field(field); ref.texture_view()[:] = field
        with np.errstate(all="ignore"):
            for s in range(steps):
                sim.tick(tick); ref.step(tick, stable_sort=stable)
                assert_particles_equal(sim.download_particles(), ref.particles(), f"fuzz case {case} mode {mode} step {s}")
                assert np.array_equal(sim.download_start_indices(), ref.start_indices()), f"fuzz case {case}: start_indices"
        cells, cnt = np.unique(ref.particles()["grid"], return_counts=True)
        sim.close(); ref.close()
    if (case - first) % 10 == 9:
        print(f"cases {first}..{case} ok (last: n={n}, max particles/cell {cnt.max()}) {time.time()-t0:.0f}s", flush=True)
print("fuzz ok:", cases, "cases x 2 sort modes")

# ---- 3D: random sides, jitter, velocities, compressions, coincident groups --------------------------------------
from test_3d import _assert_equal3
t0 = time.time()
cases3 = cases // 3
for case in range(first, first + cases3):
    rng = np.random.default_rng(9000 + case)
    side = int(rng.integers(2, 34))
    n = side ** 3
    st, off, tick = fs.dam_break_3d(n)
    sim = fs.FluidSimulation3D(st, device=0, initial_offset=off)
    ref = orc.OracleSim3D(st, off)
    p = ref.particles()
    centre = p["position"].mean(axis=0)
    squeeze = float(rng.choice([1.0, 1.0, 0.5, 0.25]))      # < 1: denser than the lattice -> long rows, list path
    p["position"] = ((p["position"] - centre) * np.float32(squeeze) + centre).astype(np.float32)
    p["position"] += rng.uniform(-0.03, 0.03, size=(n, 3)).astype(np.float32)
    if case % 4 == 1 and n > 20:
        p["position"][1:int(rng.integers(2, 6))] = p["position"][0]
    p["predicted_position"] = p["position"]
    p["velocity"] = (rng.standard_normal((n, 3)) * float(rng.choice([0.0, 1e-6, 1.0, 100.0]))).astype(np.float32)
    ref.set_particles(p); sim.upload_particles(p)
    with np.errstate(all="ignore"):
        for s in range(3):
            sim.tick(tick); ref.step(tick)
            _assert_equal3(sim.download_particles(), ref.particles(), f"fuzz3d case {case} step {s}")
    if (case - first) % 10 == 9:
        print(f"3D cases {first}..{case} ok (last: side {side}, squeeze {squeeze}) {time.time()-t0:.0f}s", flush=True)
print("fuzz3d ok:", cases3, "cases")
