"""What does k_force_general have to do in a small scene?  Per sampled step: the waves that leave the mask path (a sweep row longer
than 32 candidates), and per such wave the candidates and in-radius neighbours of its heaviest lane (= the wave's latency, the
kernel's tail at 1 M particles).   python tools/general_work_stats.py [n=1048576] [steps ...]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from scipy.spatial import cKDTree
import gpu_fluid_simulation_amd as g
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
targets = [int(x) for x in sys.argv[2:]] or [10, 35, 60, 85, 110]
st, off, tick = g.dam_break_2d(n)
sim = g.FluidSimulation(st, device=0, initial_offset=off)
gw, gh = sim.grid_dims
done = 0
for target in targets:
    while done < target:
        sim.tick(tick); done += 1
    a = sim.download_particles()
    key = a["grid"].astype(np.int64)
    cnt = np.bincount(key, minlength=gw * gh + 2 * gw + 4).astype(np.int64)
    c = np.concatenate([[0], np.cumsum(cnt)])
    def row(idc):
        lo = np.clip(idc - 1, 0, len(cnt)); hi = np.clip(idc + 2, 0, len(cnt))
        return c[hi] - c[lo]
    r0, r1, r2 = row(key - gw), row(key), row(key + gw)
    longest = np.maximum(np.maximum(r0, r1), r2)
    cand = r0 + r1 + r2
    # blocks whose sweep rows do not fit the force pass's LDS tile (544 candidates per row): as one range per row, and as the
    # two segments of fs_device.h TileSegments (lanes of the first particle's cell row / the others)
    nbk = n // 256
    kb = key[: nbk * 256].reshape(nbk, 256)
    rowid = kb // gw
    def ext(idc, sel):          # [min lo, max hi) over the selected lanes, per block
        lo = c[np.clip(idc - 1, 0, len(cnt))]; hi = c[np.clip(idc + 2, 0, len(cnt))]
        lo = np.where(sel & (hi > lo), lo, np.iinfo(np.int64).max); hi = np.where(sel & (hi > lo), hi, 0)
        e = hi.max(axis=1) - lo.min(axis=1)
        return np.maximum(e, 0)
    allsel = np.ones_like(kb, dtype=bool); inA = rowid == rowid[:, :1]
    one = np.stack([ext(kb + d * gw, allsel) for d in (-1, 0, 1)]).max(axis=0)
    two = np.stack([ext(kb + d * gw, inA) + ext(kb + d * gw, ~inA) for d in (-1, 0, 1)]).max(axis=0)
    straddle = (rowid[:, 0] != rowid[:, -1])
    print(f"    blocks {nbk}: straddling a cell-row end {int(straddle.sum())}; unfit (> 544) as one range {int((one > 544).sum())}, as two segments {int((two > 544).sum())}"
          f"; two-segment extents of the straddling blocks: median {np.median(two[straddle]) if straddle.any() else 0:.0f} max {two[straddle].max() if straddle.any() else 0}; "
          f"largest extent of the other blocks {one[~straddle].max()}", flush=True)
    tree = cKDTree(a["predicted_position"].astype(np.float64))
    m = (n // 64) * 64
    lane_long = (longest > 32)[:m].reshape(-1, 64)
    waves = lane_long.any(axis=1)
    if not waves.any():
        print(f"step {done}: no deferred wave", flush=True); continue
    idx = np.nonzero(np.repeat(waves, 64))[0]
    nb = np.array(tree.query_ball_point(a["predicted_position"][idx].astype(np.float64), float(st.smoothing_radius), return_length=True)) - 1
    nbw = nb.reshape(-1, 64); cw = cand[:m].reshape(-1, 64)[waves]
    blocks = waves[: (len(waves) // 4) * 4].reshape(-1, 4)
    print(f"step {done}: deferred waves {int(waves.sum())} of {len(waves)} in {int(blocks.any(axis=1).sum())} blocks "
          f"(waves per such block {blocks.sum(axis=1)[blocks.any(axis=1)].mean():.2f}); heaviest lane per wave: candidates mean {cw.max(axis=1).mean():.0f} "
          f"p90 {np.percentile(cw.max(axis=1), 90):.0f} max {cw.max()}; in-radius mean {nbw.max(axis=1).mean():.0f} p90 {np.percentile(nbw.max(axis=1), 90):.0f} "
          f"max {nbw.max()}; all lanes of these waves: in-radius mean {nbw.mean():.1f}; lane utilisation (mean / wave max) {(nbw.mean(axis=1) / np.maximum(nbw.max(axis=1), 1)).mean():.2f}", flush=True)
