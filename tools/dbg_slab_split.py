import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from scipy.spatial import cKDTree
import gpu_fluid_simulation_amd as g
from test_multi_gpu import InProcessSlabs
n = 4096
st, off, tick = g.dam_break_2d(n)
slabs = InProcessSlabs(g, st, off, 2, cap=n + 4 * 2048, recv=2048)
single = g.FluidSimulation(st, device=0, initial_offset=off, ref_quirks=False)
single.upload_particles(slabs.initial)
for s in range(6):
    slabs.step(tick); single.tick(tick)
    a, b = slabs.owned(), single.download_particles()
    tree = cKDTree(b["predicted_position"].astype(np.float64))
    d, idx = tree.query(a["predicted_position"].astype(np.float64))
    u, c = np.unique(idx, return_counts=True)
    print("step", s + 1, "n", len(a), "unique matches", len(u), "max d", d.max(), "dups", int((c > 1).sum()))
    if (c > 1).any() or d.max() > 1e-4:
        bad = np.nonzero(d > 1e-4)[0][:6]
        for k in bad:
            print("   got", a[k], " nearest want", b[idx[k]])
        # which single-engine particles have no partner
        missing = np.setdiff1d(np.arange(len(b)), u)[:6]
        for k in missing: print("   missing want", b[k])
        break
