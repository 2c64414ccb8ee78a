"""Per-pass times of the 16M dam break at several simulated times (how the nearly-sorted
assumption of the sort's exact skipping holds as the fluid gets disordered)."""
import os, sys
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
st, off, tick = g.dam_break_2d(n)
sim = g.FluidSimulation(st, device=0, initial_offset=off)
done = 0
for target in (20, 500, 1500, 3000, 5000):
    while done < target:
        sim.tick(tick); done += 1
    sim.sync(); sim.profile(True); sim.profile_read(True)
    ms = sim.timed_steps(tick, 20); done += 20
    p, k = sim.profile_read(True); sim.profile(False)
    print(f"step {done:5d}: {ms/20:.3f} ms/step  {n/(ms/20*1e-3)/1e6:8.1f} M p-s/s ", {a: round(b/20, 3) for a, b in p.items()}, flush=True)
