import os, sys, time
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
n = 1 << 24
st, off, tick = g.dam_break_2d(n)
def window(label, preheat):
    if preheat:
        s0 = g.FluidSimulation(st, device=0, initial_offset=off)
        for _ in range(preheat): s0.tick(tick)
        s0.sync(); s0.close()
    sim = g.FluidSimulation(st, device=0, initial_offset=off)
    for _ in range(10): sim.tick(tick)
    sim.sync()
    ms = sim.timed_steps(tick, 100)
    print(label, round(ms / 100, 4), "ms/step", round(n / (ms / 100 * 1e-3) / 1e6), "M p-s/s", flush=True)
    sim.close()
window("cold (10 warm-up steps)", 0)
window("after 300 throw-away steps", 300)
window("again", 0)
