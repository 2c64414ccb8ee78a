import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import gpu_fluid_simulation_amd as g
from oracle import oracle as O
O.set_threads(16)
n3 = 48 ** 3
st, off, tick = g.dam_break_3d(n3)
sim = g.FluidSimulation3D(st, device=0, initial_offset=off); ref = O.OracleSim3D(st, off)
F = ("position", "predicted_position", "velocity", "density")
ok = True
for s in range(1, 241):
    sim.tick(tick); ref.step(tick)
    if s % 40 == 0:
        a, b = sim.download_particles(), ref.particles()
        good = all(np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)) for f in F) and np.array_equal(a["grid"], b["grid"])
        cells, cnt = np.unique(b["grid"], return_counts=True)
        print(f"3D n={n3} step {s}: bit-exact={good} max/cell={cnt.max()}", flush=True)
        ok &= good
assert ok
print("3D long validation ok")
