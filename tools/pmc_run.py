"""Workload for `rocprofv3 --pmc ... -- python3 tools/pmc_run.py [steps]`: 16M dam break, `steps` ticks."""
import os, sys
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
st, off, tick = g.dam_break_2d(1 << 24)
sim = g.FluidSimulation(st, device=0, initial_offset=off)
for _ in range(steps):
    sim.tick(tick)
sim.sync()
print("done", steps, flush=True)
