"""Row statistics of the 8 M 3D scene after N steps: per 256-particle block of the sorted order, the extent of its nine
sweep rows (what k3_density / k3_force stage per row, against TILE3), and per wave the longest row (masks need <= 64).
  python tools/rows3d_stats.py [steps ...]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import gpu_fluid_simulation_amd as g

steps_list = [int(a) for a in sys.argv[1:]] or [10, 60, 110]
st, off, tick = g.dam_break_3d(200 ** 3)
sim = g.FluidSimulation3D(st, device=0, initial_offset=off)
gw, gh, gd = sim.grid_dims
done = 0
for target in steps_list:
    while done < target:
        sim.tick(tick); done += 1
    p = sim.download_particles()
    key = p["grid"].astype(np.int64)
    assert np.all(np.diff(key) >= 0)
    n = key.shape[0]
    ncell = gw * gh * gd
    cs = np.searchsorted(key, np.arange(ncell + 4), side="left")
    nb = (n + 255) // 256
    first = key[np.arange(nb) * 256]
    last = key[np.minimum(np.arange(nb) * 256 + 255, n - 1)]
    ext_max = np.zeros(nb, dtype=np.int64)
    wfirst = key[np.arange((n + 63) // 64) * 64]
    wlast = key[np.minimum(np.arange((n + 63) // 64) * 64 + 63, n - 1)]
    unfit_planes = 0
    for oz in (-1, 0, 1):
        plane_ext = np.zeros(nb, dtype=np.int64)
        for oy in (-1, 0, 1):
            o = (oz * gh + oy) * gw
            lo = cs[np.clip(first + o - 1, 0, ncell)]
            hi = cs[np.clip(last + o + 2, 0, ncell)]
            plane_ext = np.maximum(plane_ext, hi - lo)
        unfit_planes += int((plane_ext > 400).sum())
        ext_max = np.maximum(ext_max, plane_ext)
    # longest 3-cell row any particle of a wave sees (upper bound via per-cell counts)
    cnt = np.diff(cs[:ncell + 1])
    row3 = cnt.copy(); row3[1:] += cnt[:-1]; row3[:-1] += cnt[1:]
    per_particle = np.zeros(n, dtype=np.int64)
    for oz in (-1, 0, 1):
        for oy in (-1, 0, 1):
            o = (oz * gh + oy) * gw
            per_particle = np.maximum(per_particle, row3[np.clip(key + o, 0, ncell - 1)])
    wmax = np.maximum.reduceat(per_particle, np.arange(0, n, 64))
    q = np.percentile(ext_max, [50, 90, 99, 100])
    print(f"step {done}: blocks {nb}; row extent per block p50/p90/p99/max = {q}; planes over 400: {unfit_planes} of {3 * nb} "
          f"({100.0 * unfit_planes / (3 * nb):.2f} %); over 384: {int((ext_max > 384).sum())} blocks; "
          f"waves with a row > 64: {int((wmax > 64).sum())} of {wmax.shape[0]} ({100.0 * (wmax > 64).mean():.2f} %), > 128: {100.0 * (wmax > 128).mean():.2f} %, > 192: {100.0 * (wmax > 192).mean():.2f} %; "
          f"max particles per cell {int(cnt.max())}; mean neighbours-in-3x3x3 candidates {float(per_particle.mean()) * 3:.0f}")
