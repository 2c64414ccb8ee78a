import sys, os
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import _abi
variant = sys.argv[1]
if variant != "default":
    _abi._lib = _abi.load_library(os.path.join("gpu-fluid-simulation_amd", variant))
st, off, tick = g.dam_break_3d(200 ** 3)
sim = g.FluidSimulation3D(st, device=0, initial_offset=off)
for _ in range(10): sim.tick(tick)
sim.sync(); sim.profile(True); sim.profile_read(True)
ms = sim.timed_steps(tick, 40)
p, k = sim.profile_read(True)
print(variant, round(ms/40, 4), {a: round(b/40, 4) for a, b in p.items()})
