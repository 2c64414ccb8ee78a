"""Per-pass times of the 16M dam break in a math / sort mode: python tools/ab_mode.py <strict|ulp|tol> [bitonic|counting] [warm] [steps] [lib]"""
import sys, os
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import _abi
mode = sys.argv[1] if len(sys.argv) > 1 else "strict"
sort = sys.argv[2] if len(sys.argv) > 2 else "bitonic"
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 10
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
if len(sys.argv) > 5 and sys.argv[5] != "default":
    _abi._lib = _abi.load_library(os.path.join("gpu-fluid-simulation_amd", sys.argv[5]))
mm = {"strict": g.FS_MATH_IEEE, "ulp": g.FS_MATH_WGSL_ULP, "tol": g.FS_MATH_TOLERANCE}[mode]
sm = g.FS_SORT_BITONIC if sort == "bitonic" else g.FS_SORT_COUNTING
n = 1 << 24
st, off, tick = g.dam_break_2d(n)
sim = g.FluidSimulation(st, device=0, initial_offset=off, math_mode=mm, sort_mode=sm)
for _ in range(warm): sim.tick(tick)
sim.sync(); sim.profile(True); sim.profile_read(True)
ms = sim.timed_steps(tick, steps)
p, k = sim.profile_read(True)
print(mode, sort, f"steps {warm}-{warm+steps}", round(ms / steps, 4), {a: round(b / steps, 4) for a, b in p.items()}, sim.sort_plan(), flush=True)
