"""ms/step over the official window (10 warm-up + 100 steps) for the strict default mode."""
import os, sys
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
n = 1 << 24
st, off, tick = g.dam_break_2d(n)
sim = g.FluidSimulation(st, device=0, initial_offset=off)
for _ in range(10): sim.tick(tick)
sim.sync(); sim.profile(True); sim.profile_read(True)
ms = sim.timed_steps(tick, 100) / 100
p, k = sim.profile_read(True)
print(os.environ.get("TAG", ""), round(ms, 4), {a: round(b / 100, 4) for a, b in p.items()})
