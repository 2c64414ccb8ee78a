"""One-off API-sequence fuzz: the same random sequence of calls — ticks with changing tick settings, particle
uploads mid-run, force-field uploads, downloads — on the GPU engine and on the oracle; every download must be
bit-identical (particles and start_indices), in both sort modes.  python tools/fuzz_api.py [first] [cases]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import gpu_fluid_simulation_amd as fs
from oracle import oracle as orc
from test_parity_gpu import assert_particles_equal

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = time.time()
for case in range(first, first + cases):
    rng = np.random.default_rng(11000 + case)
    n = int(rng.integers(2, 20000))
    h = float(rng.choice([0.1, 0.2, 0.5]))
    spacing = float(h * rng.uniform(0.25, 0.9))
    side = np.sqrt(n) * spacing
    size = (float(side * rng.uniform(1.2, 2.5) + 4 * h), float(side * rng.uniform(1.2, 2.5) + 4 * h))
    tex = (int(rng.choice([32, 64, 256])), int(rng.choice([32, 128])))
    st = fs.SimulationSettings(n, spacing, h, size, tex)
    mode = fs.FS_SORT_BITONIC if case % 2 == 0 else fs.FS_SORT_COUNTING
    quirks = bool(case % 3)
    sim = fs.FluidSimulation(st, device=0, sort_mode=mode, ref_quirks=quirks)
    ref = orc.OracleSim(st, (0.0, 0.0), ref_quirks=quirks)
    def new_tick():
        return fs.default_tick_settings(
            delta=float(rng.choice([1 / 240, 1 / 120, 1 / 60])), gravity=(float(rng.uniform(-5, 5)), float(rng.uniform(-10, 10))),
            mass=float(rng.uniform(0.5, 2.0)), pressure_constant=float(rng.uniform(5, 100)),
            rest_density=float(rng.choice([0.0, 1.0, 20.0])), damping_factor=float(rng.uniform(0.0, 0.9)),
            viscosity_coefficient=float(rng.choice([0.0, 5.0, 25.0])), mouse_state=int(rng.choice([0, 0, 1, -1])),
            mouse_pos=(float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))), mouse_force_radius=float(rng.uniform(0.5, 5)))
    tick = new_tick()
    log = []
    with np.errstate(all="ignore"):
        for op_i in range(int(rng.integers(6, 16))):
            op = rng.choice(["tick", "tick", "tick", "settings", "upload", "field", "check"])
            log.append(str(op))
            if op == "tick":
                for _ in range(int(rng.integers(1, 5))):
                    sim.tick(tick); ref.step(tick, stable_sort=(mode == fs.FS_SORT_COUNTING))
            elif op == "settings":
                tick = new_tick()
            elif op == "upload":
                p = ref.particles()
                p["position"] = (p["position"] + rng.uniform(-0.2, 0.2, size=(n, 2)).astype(np.float32) * np.float32(spacing))
                p["predicted_position"] = p["position"]
                p["velocity"] = (rng.standard_normal((n, 2)) * float(rng.choice([0.0, 1.0, 20.0]))).astype(np.float32)
                ref.set_particles(p); sim.upload_particles(p)
            elif op == "field":
                field = np.zeros((tex[1], tex[0], 2), dtype=np.float32)
                if rng.random() < 0.7:
                    field[tex[1] // 4: tex[1] // 2, tex[0] // 4: 3 * tex[0] // 4] = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)))
                sim.upload_force_field(field); ref.texture_view()[:] = field
            else:
                assert_particles_equal(sim.download_particles(), ref.particles(), f"api fuzz case {case} after {log}")
                assert np.array_equal(sim.download_start_indices(), ref.start_indices()), f"case {case}: start_indices after {log}"
        assert_particles_equal(sim.download_particles(), ref.particles(), f"api fuzz case {case} end after {log}")
        assert np.array_equal(sim.download_start_indices(), ref.start_indices()), f"case {case}: start_indices at end"
        assert sim.tick_count == ref.tick_count
    sim.close(); ref.close()
    if (case - first) % 10 == 9:
        print(f"cases {first}..{case} ok ({time.time()-t0:.0f}s)", flush=True)
print("api fuzz ok:", cases, "cases")
