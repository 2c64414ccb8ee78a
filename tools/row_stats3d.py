import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import gpu_fluid_simulation_amd as g
st, off, tick = g.dam_break_3d(200 ** 3)
sim = g.FluidSimulation3D(st, device=0, initial_offset=off)
gw, gh, gd = sim.grid_dims
done = 0
sim.profile(True)
for target in (10, 50, 80, 110):
    sim.profile_read(True)
    t0 = done
    while done < target:
        sim.tick(tick); done += 1
    sim.sync()
    p, k = sim.profile_read(True)
    key = sim.download_particles()["grid"].astype(np.int64)
    ncell = gw * gh * gd
    cnt = np.bincount(key, minlength=ncell + 4).astype(np.int64)
    c = np.concatenate([[0], np.cumsum(cnt)])
    row = c[np.clip(key + 2, 0, len(cnt))] - c[np.clip(key - 1, 0, len(cnt))]
    blocks = row[: (len(row) // 256) * 256].reshape(-1, 256)
    # tile need per block ~ span of candidate indices of one row over the block
    lo = c[np.clip(key - 1, 0, len(cnt))]; hi = c[np.clip(key + 2, 0, len(cnt))]
    span = hi[: (len(row) // 256) * 256].reshape(-1, 256).max(1) - lo[: (len(row) // 256) * 256].reshape(-1, 256).min(1)
    print(f"steps {t0}-{done}: force {p['force']/(done-t0):.3f} ms density {p['density']/(done-t0):.3f}; per cell mean {cnt[cnt>0].mean():.2f} max {cnt.max()}; "
          f"own-row len mean {row.mean():.1f} p99 {np.percentile(row,99):.0f} max {row.max()}; blocks with a row>32: {(blocks.max(1)>32).mean()*100:.1f}% >64: {(blocks.max(1)>64).mean()*100:.1f}% >96: {(blocks.max(1)>96).mean()*100:.1f}% >128: {(blocks.max(1)>128).mean()*100:.1f}% >192: {(blocks.max(1)>192).mean()*100:.1f}%; waves(64) >64: {(row[:(len(row)//64)*64].reshape(-1,64).max(1)>64).mean()*100:.1f}% >128: {(row[:(len(row)//64)*64].reshape(-1,64).max(1)>128).mean()*100:.1f}%; "
          f"own-row tile span p50 {np.percentile(span,50):.0f} p99 {np.percentile(span,99):.0f} (>384: {(span>384).mean()*100:.1f}%)", flush=True)
