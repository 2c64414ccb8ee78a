"""Workload for `rocprofv3 --pmc ... -- python3 tools/pmc_run.py <scene> [warm] [steps] [mode]`.
scene: 2d (16M dam break) | 3d (8M dam break) | 2d1m.  The library must be built already: nothing is compiled
here (the profiler's preload has initialised the GPU; exec'ing compilers from this process is not allowed).
mode: strict (default) | counting | ulp | tol (FS_MATH_TOLERANCE)."""
import os, sys
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
g.load_library()
scene = sys.argv[1] if len(sys.argv) > 1 else "2d"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 10
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mode = sys.argv[4] if len(sys.argv) > 4 else "strict"
if scene == "3d":
    st, off, tick = g.dam_break_3d(200 ** 3)
    sim = g.FluidSimulation3D(st, device=0, initial_offset=off, math_mode=g.FS_MATH_TOLERANCE if mode == "tol" else g.FS_MATH_IEEE)
else:
    st, off, tick = g.dam_break_2d(1 << 20 if scene == "2d1m" else 1 << 24)
    kw = {}
    if mode == "counting": kw["sort_mode"] = g.FS_SORT_COUNTING
    if mode == "ulp": kw["math_mode"] = g.FS_MATH_WGSL_ULP
    if mode == "tol": kw["math_mode"] = g.FS_MATH_TOLERANCE
    sim = g.FluidSimulation(st, device=0, initial_offset=off, **kw)
for _ in range(warm + steps):
    sim.tick(tick)
sim.sync()
print("done", scene, warm, steps, mode, flush=True)
