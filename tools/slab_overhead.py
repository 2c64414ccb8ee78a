"""Single-rank slab engine vs plain engine on one GPU (same scene): what the fixed-capacity,
device-resident-count design costs before any communication."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import multi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
st, off, tick = g.dam_break_2d(n)
hist, gw = multi.lattice_histogram(g, st, off)
gh = int(np.ceil(np.float32(st.size.y) / np.float32(st.smoothing_radius))) + 2
bounds = multi.trim_outer_edges(multi.partition_columns(hist, world), hist, multi.default_trim_margin())
cap, recv = multi.slab_capacities(n, world, gh)
lat = g.reference_lattice(st, off)
cols = multi.global_columns(lat["position"][:, 0], st.size.x, st.smoothing_radius)
# emulate rank `world//2` of `world` ranks WITHOUT neighbours (no exchange): pure local cost
r = int(sys.argv[3]) if len(sys.argv) > 3 else world // 2
sim = g.SlabSimulation(st, bounds[r], bounds[r + 1], False, False, cap, recv, max_cols=min(gw, 2 * (bounds[r + 1] - bounds[r]) + 64))
own = lat[(cols >= bounds[r]) & (cols < bounds[r + 1])]
sim.upload_owned(own)
print(f"rank {r}/{world}: owned {own.shape[0]} capacity {cap} recv {recv} cols {bounds[r+1]-bounds[r]}")
for _ in range(10):
    sim.pack(tick, None, None); sim.step(None, None)
sim.sync(); sim.profile(True); sim.profile_read(True)
t0 = time.perf_counter()
K = 50
for _ in range(K):
    sim.pack(tick, None, None); sim.step(None, None)
sim.sync()
el = (time.perf_counter() - t0) / K * 1e3
p, k = sim.profile_read(True)
print(f"slab local step: {el:.3f} ms  ({own.shape[0]/(el*1e-3)/1e6:.0f} M p-s/s on this rank)", {a: round(b / K, 3) for a, b in p.items()}, sim.counters())
