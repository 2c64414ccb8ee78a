"""One rank of an N-way split of the dam break on ONE GPU: what the slab engine costs per step

  python3 tools/slab_overhead.py N WORLD [RANK] [--neighbours] [--mode edge|strips|serial] [--steps K] [--boundary Z] [--no-exchange]

without neighbours (default): the pure local cost of the rank (no exchange, nothing that depends on one);
--neighbours: ranks RANK-1, RANK, RANK+1 of the split live on this GPU and run WARMUP steps together (messages copied through
the host).  Then the two neighbours' outgoing messages are FROZEN — their migrants removed, so that the middle rank does not
receive the same particles every step — and the middle rank runs alone and free of host synchronisation:
    pack -> exchange -> step,   K times, one synchronisation at the end
where the exchange is the C ABI's RCCL binding on a single-rank communicator: fs_slab_exchange with left_rank = right_rank =
own rank copies "send_left" into recv_left and "send_right" into recv_right, and is handed the frozen neighbour messages as
the sources — the rank receives halo messages of the real size through ncclSend / ncclRecv, on its exchange stream in the two
overlapped modes and on the simulation's stream with --mode serial.  (Its own outgoing messages go nowhere.  The frozen halo
is a few steps stale by the end of the window; the timings are what this tool is for.)
Reported: ms per step by the host clock over the free-running window, then a second window with an event at every pass
boundary of the rank's stream (the events themselves cost ~5 us per step).
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import multi

ap = argparse.ArgumentParser()
ap.add_argument("n", type=int, nargs="?", default=1 << 24)
ap.add_argument("world", type=int, nargs="?", default=1)
ap.add_argument("rank", type=int, nargs="?", default=None)
ap.add_argument("--neighbours", action="store_true")
ap.add_argument("--mode", choices=["edge", "strips", "serial"], default="edge")
ap.add_argument("--no-exchange", action="store_true", help="with --neighbours: hand the frozen buffers over directly, no RCCL call (not with --mode edge)")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--boundary", type=int, default=0)
ap.add_argument("--lib", default=None, help="a variant build of the library inside the package directory")
a = ap.parse_args()
if a.lib:
    from gpu_fluid_simulation_amd import _abi
    _abi._lib = _abi.load_library(os.path.join("gpu-fluid-simulation_amd", a.lib))

n, world = a.n, a.world
st, off, tick = g.dam_break_2d(n)
hist, gw = multi.lattice_histogram(g, st, off)
gh = int(np.ceil(np.float32(st.size.y) / np.float32(st.smoothing_radius))) + 2
bounds = multi.trim_outer_edges(multi.partition_columns(hist, world), hist, multi.default_trim_margin())
cap, recv = multi.slab_capacities(n, world, gh)
lat = g.reference_lattice(st, off)
cols = multi.global_columns(lat["position"][:, 0], st.size.x, st.smoothing_radius)
r = a.rank if a.rank is not None else world // 2
nb = a.neighbours
if nb and not (0 < r < world - 1):
    sys.exit("--neighbours needs a rank with two neighbours")


def make(rank, has_l, has_r):
    s = g.SlabSimulation(st, bounds[rank], bounds[rank + 1], has_l, has_r, cap, recv,
                         max_cols=min(gw, 2 * (bounds[rank + 1] - bounds[rank]) + 64), serial=a.mode == "serial",
                         strips=a.mode == "strips")
    if a.boundary and s.overlapped:
        s.set_boundary_cols(a.boundary)
    s.upload_owned(lat[(cols >= bounds[rank]) & (cols < bounds[rank + 1])])
    return s


sim = make(r, nb, nb)
own_n = int(((cols >= bounds[r]) & (cols < bounds[r + 1])).sum())
print(f"rank {r}/{world}: owned {own_n} capacity {cap} recv {recv} cols {bounds[r+1]-bounds[r]} mode {a.mode} ({sim.step_mode}) "
      f"boundary_cols {sim.boundary_cols} neighbours {nb}")

lib = g.load_library()
P = lambda b: C.c_void_p(b.device_ptr) if b is not None else None
comm = None
if nb:
    L, R = make(r - 1, False, True), make(r + 1, True, False)       # their outer edges are open: irrelevant to the middle rank's cost
    mb = sim.message_bytes
    B = {k: g.ResizableBuffer(k, np.uint8, mb) for k in ("m_sl", "m_sr", "m_rl", "m_rr", "l_sr", "l_rr", "r_sl", "r_rl", "fl", "fr")}
    for _ in range(a.warmup):       # the three ranks together, messages copied through the host
        L.pack(tick, None, P(B["l_sr"])); sim.pack(tick, P(B["m_sl"]), P(B["m_sr"])); R.pack(tick, P(B["r_sl"]), None)
        for s_ in (L, sim, R):
            s_.wait_packed()
        B["m_rl"].write(0, B["l_sr"].read()); B["m_rr"].write(0, B["r_sl"].read())
        B["l_rr"].write(0, B["m_sl"].read()); B["r_rl"].write(0, B["m_sr"].read())
        L.step(None, P(B["l_rr"])); sim.step(P(B["m_rl"]), P(B["m_rr"])); R.step(P(B["r_rl"]), None)
    for s_ in (L, sim, R):
        s_.sync()
    # freeze the neighbours' outgoing messages without their migrants (records whose predicted column the middle rank owns)
    for src, dst, keep in (("l_sr", "fl", lambda c: c < bounds[r]), ("r_sl", "fr", lambda c: c >= bounds[r + 1])):
        raw = B[src].read()
        hdr = raw[:16].view(np.uint32).copy()
        rec = raw[16:16 + 16 * int(hdr[0])].view(np.float32).reshape(-1, 4)
        px = rec[:, 0] + rec[:, 2] * np.float32(tick.delta)
        sel = rec[keep(multi.global_columns(px, st.size.x, st.smoothing_radius))]
        out = np.zeros(mb, dtype=np.uint8)
        hdr[0], hdr[1] = sel.shape[0], 0
        out[:16] = hdr.view(np.uint8)
        out[16:16 + 16 * sel.shape[0]] = np.ascontiguousarray(sel).view(np.uint8).reshape(-1)
        B[dst].write(0, out)
        print(f"frozen {dst}: {sel.shape[0]} ghost records of {int(rec.shape[0])}")
    L.close(); R.close()
    if not a.no_exchange:
        idb = (C.c_uint8 * 128)()
        g._check(lib, lib.fs_comm_unique_id(idb))
        comm = C.c_void_p()
        g._check(lib, lib.fs_comm_init(0, 0, 1, idb, C.byref(comm)))
    elif a.mode == "edge":
        pass    # the frozen buffers are never refilled by anybody: sharing them is fine


def step():
    if not nb:
        sim.pack(tick, None, None)
        sim.step(None, None)
        return
    sim.pack(tick, P(B["m_sl"]), P(B["m_sr"]))
    if comm is not None:    # "send_left" := the (frozen) LEFT neighbour's outgoing message, "send_right" := the right neighbour's
        g._check(lib, lib.fs_slab_exchange(sim._h, comm, 0, 0, P(B["fl"]), P(B["fr"]), P(B["m_rl"]), P(B["m_rr"])))
        sim.step(P(B["m_rl"]), P(B["m_rr"]))
    else:
        sim.step(P(B["fl"]), P(B["fr"]))


if not nb:
    for _ in range(a.warmup):
        step()
else:
    for _ in range(3):
        step()
sim.sync()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
host = (time.perf_counter() - t0) / a.steps * 1e3      # what the host needs to ENQUEUE a step (the loop never waits for the GPU)
sim.sync()
el = (time.perf_counter() - t0) / a.steps * 1e3
print(f"host enqueue time per step: {host:.4f} ms")
sim.profile(True); sim.profile_read(True)
for _ in range(a.steps):
    step()
sim.sync()
p, k = sim.profile_read(True)
pp = {x: round(y / max(k, 1), 4) for x, y in p.items()}
print(f"slab local step: {el:.4f} ms by the host clock ({own_n/(el*1e-3)/1e6:.0f} M p-s/s on this rank); by events "
      f"{sum(pp.values()):.4f} ms", pp, sim.counters())
if comm is not None:
    lib.fs_comm_destroy(comm)
