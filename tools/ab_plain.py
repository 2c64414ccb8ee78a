"""Step time of a 2D dam break WITHOUT per-pass events (what bench.py's value is timed like):
  python tools/ab_plain.py <default|variant.so> <n> [warm] [steps]"""
import sys, os
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import _abi
variant = sys.argv[1]; n = int(sys.argv[2])
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 10
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
if variant != "default":
    _abi._lib = _abi.load_library(os.path.join("gpu-fluid-simulation_amd", variant))
st, off, tick = g.dam_break_2d(n)
sim = g.FluidSimulation(st, device=0, initial_offset=off)
for _ in range(warm): sim.tick(tick)
sim.sync()
ms = sim.timed_steps(tick, steps)
print(variant, n, f"steps {warm}-{warm + steps}", round(ms / steps, 4), flush=True)
