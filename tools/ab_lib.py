"""Per-pass times of a 2D dam break of n particles with a variant build of the library (a file name inside the package
directory, e.g. one built from another commit: build.build(out=...)) or `default`:
  python tools/ab_lib.py <lib.so|default> <n> [warm] [steps] [bitonic|counting]"""
import sys, os
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import _abi
variant = sys.argv[1]
if variant != "default":
    _abi._lib = _abi.load_library(os.path.join("gpu-fluid-simulation_amd", variant))
n = int(sys.argv[2]); warm = int(sys.argv[3]) if len(sys.argv) > 3 else 10; steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
sort = {"bitonic": g.FS_SORT_BITONIC, "counting": g.FS_SORT_COUNTING}[sys.argv[5] if len(sys.argv) > 5 else "bitonic"]
st, off, tick = g.dam_break_2d(n)
sim = g.FluidSimulation(st, device=0, initial_offset=off, sort_mode=sort)
for _ in range(warm): sim.tick(tick)
sim.sync()
ms0 = sim.timed_steps(tick, steps)          # free-running (one event pair around the window)
sim.close()
sim = g.FluidSimulation(st, device=0, initial_offset=off, sort_mode=sort)
for _ in range(warm): sim.tick(tick)
sim.sync(); sim.profile(True); sim.profile_read(True)
ms = sim.timed_steps(tick, steps)
p, k = sim.profile_read(True)
print(variant, n, sort, f"steps {warm}-{warm+steps}: {ms0 / steps:.4f} ms/step; with pass events {ms / steps:.4f}", {a: round(b / steps, 4) for a, b in p.items()}, flush=True)
