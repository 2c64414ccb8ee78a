import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import gpu_fluid_simulation_amd as g
n = 1 << 24
st, off, tick = g.dam_break_2d(n)
sim = g.FluidSimulation(st, device=0, initial_offset=off)
done = 0
for target in (0, 10, 30, 60, 110, 160):
    while done < target:
        sim.tick(tick); done += 1
    sim.sync(); sim.profile(True); sim.profile_read(True)
    ms = sim.timed_steps(tick, 10); done += 10
    p, k = sim.profile_read(True); sim.profile(False)
    print(f"steps {done-10:4d}-{done:4d}: {ms/10:.3f} ms/step ", {a: round(b/10, 3) for a, b in p.items()}, flush=True)
pp = sim.download_particles()
print("density max/mean", pp["density"].max(), pp["density"].mean(), " |v| max", np.hypot(*pp["velocity"].T).max())
cells, cnt = np.unique(pp["grid"], return_counts=True)
print("particles per occupied cell: mean", cnt.mean(), "max", cnt.max(), "p99", np.percentile(cnt, 99))
