"""Per-pass times of a 2D dam break of n particles: python tools/ab_n.py <n> [warm] [steps]  (env FS_SORT_GB etc. apply)"""
import sys, os
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
n = int(sys.argv[1]); warm = int(sys.argv[2]) if len(sys.argv) > 2 else 10; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
st, off, tick = g.dam_break_2d(n)
sim = g.FluidSimulation(st, device=0, initial_offset=off)
for _ in range(warm): sim.tick(tick)
sim.sync(); sim.profile(True); sim.profile_read(True)
ms = sim.timed_steps(tick, steps)
p, k = sim.profile_read(True)
print(n, os.environ.get("FS_SORT_GB", "auto"), f"steps {warm}-{warm+steps}", round(ms / steps, 4), {a: round(b / steps, 4) for a, b in p.items()}, flush=True)
