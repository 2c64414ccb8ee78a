"""Protocol check at the bench's real multi-GPU geometry, on one GPU: the 16 M dam break split into `world`
column slabs (capacities and message sizes exactly as bench.py --gpus N sets them), all slabs stepped in turn on
device 0 with their messages handed over directly, re-balanced every 64 steps — counters (lost / overflow /
far_halo) and particle conservation over the 10 + 100-step window.  python tools/slab_bench_geometry.py [world] [steps]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import multi
from test_multi_gpu import InProcessSlabs
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 110
n = 1 << 24
st, off, tick = g.dam_break_2d(n)
gh = int(np.ceil(np.float32(st.size.y) / np.float32(st.smoothing_radius))) + 2
cap, recv = multi.slab_capacities(n, world, gh)
margin = multi.default_trim_margin()
slabs = InProcessSlabs(g, st, off, world, cap=cap, recv=recv, trim_margin=margin)
print(f"world {world}: capacity {cap} recv {recv} message {slabs.sims[0].message_bytes} B, outer-edge margin {margin}, columns {np.diff(slabs.bounds).tolist()}", flush=True)
t0 = time.time()
for s in range(1, steps + 1):
    slabs.step(tick)
    if s % 64 == 0:
        slabs.rebalance(2)
    if s % 10 == 0 or s == steps:
        cs = [x.counters() for x in slabs.sims]
        live = sum(c["n_live"] for c in cs)
        bad = sum(c["lost"] + c["overflow"] + c["far_halo"] for c in cs)
        print(f"step {s}: live slots {live}, violations {bad}, per-rank live {[c['n_live'] for c in cs]} ({time.time()-t0:.0f}s)", flush=True)
own = slabs.owned()
print("final columns:", np.diff(slabs.bounds).tolist())
print("owned particles:", own.shape[0], "of", n, "-> conserved" if own.shape[0] == n else "-> LOST")
assert own.shape[0] == n
slabs.assert_clean()
print("ok")
