"""Per-pass times of the 8M 3D dam break: python tools/ab_3d.py [warm] [steps] [lib]  (env FS_SORT_* selects the sort plan)"""
import os, sys
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import _abi
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 10
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lib = sys.argv[3] if len(sys.argv) > 3 else 'default'
if lib != 'default':
    _abi._lib = _abi.load_library(os.path.join('gpu-fluid-simulation_amd', lib))
st, off, tick = g.dam_break_3d(200 ** 3)
mode = g.FS_MATH_TOLERANCE if os.environ.get('FS3_TOL') else g.FS_MATH_IEEE
sim = g.FluidSimulation3D(st, device=0, initial_offset=off, math_mode=mode)
for _ in range(warm): sim.tick(tick)
sim.sync(); sim.profile(True); sim.profile_read(True)
ms = sim.timed_steps(tick, steps)
p, k = sim.profile_read(True)
print("3d", lib, "tol" if os.environ.get("FS3_TOL") else "strict", {k: v for k, v in os.environ.items() if k.startswith("FS3_")}, f"steps {warm}-{warm+steps}", round(ms / steps, 4), {a: round(b / steps, 4) for a, b in p.items()}, flush=True)
