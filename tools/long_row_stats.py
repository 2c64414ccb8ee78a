"""How often does a wave of the force pass leave the pass-mask path (a sweep row longer than 32 candidates)?
16M dam break, sampled at a few steps of the bench window.  python tools/long_row_stats.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import gpu_fluid_simulation_amd as g
n = 1 << 24
st, off, tick = g.dam_break_2d(n)
sim = g.FluidSimulation(st, device=0, initial_offset=off)
gw, gh = sim.grid_dims
done = 0
for target in (10, 60, 110, 160):
    while done < target:
        sim.tick(tick); done += 1
    key = sim.download_particles()["grid"].astype(np.int64)      # cell-sorted order = lane order
    cnt = np.bincount(key, minlength=gw * gh + 2 * gw + 4).astype(np.int64)
    c = np.concatenate([[0], np.cumsum(cnt)])
    def row(idc):                                               # candidates in cells idc-1 .. idc+1
        lo = np.clip(idc - 1, 0, len(cnt)); hi = np.clip(idc + 2, 0, len(cnt))
        return c[hi] - c[lo]
    longest = np.maximum(np.maximum(row(key - gw), row(key)), row(key + gw))
    lane_long = longest > 32
    waves = lane_long[: (n // 64) * 64].reshape(-1, 64).any(axis=1)
    print(f"step {done}: particles/cell mean {cnt[cnt > 0].mean():.2f} max {cnt.max()}; longest row: mean {longest.mean():.1f} "
          f"p99 {np.percentile(longest, 99):.0f} max {longest.max()}; lanes with a row > 32: {lane_long.mean()*100:.3f} %; "
          f"waves off the mask path: {waves.mean()*100:.3f} %", flush=True)
