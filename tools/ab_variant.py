import sys, os
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import _abi
variant = sys.argv[1]
if variant != "default":
    _abi._lib = _abi.load_library(os.path.join("gpu-fluid-simulation_amd", variant))
n = 1 << 24
st, off, tick = g.dam_break_2d(n)
sim = g.FluidSimulation(st, device=0, initial_offset=off)
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for _ in range(warm): sim.tick(tick)
sim.sync(); sim.profile(True); sim.profile_read(True)
ms = sim.timed_steps(tick, steps)
p, k = sim.profile_read(True)
print(variant, f"steps {warm}-{warm+steps}", round(ms/steps, 4), {a: round(b/steps, 4) for a, b in p.items()})
