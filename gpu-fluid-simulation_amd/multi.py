"""multi.py — multi-GPU slab driver: one process per GPU, 1-D slabs of
whole cell columns along x, one fixed-size point-to-point message per slab neighbour per
step (migrating particles + 2-column ghost halo) over RCCL (`torch.distributed`, backend
"nccl" = RCCL over xGMI).  No collective on the data path; a tiny all-reduce of the column
histogram every `rebalance_every` steps moves the slab boundaries.

The reference is single-device (src/renderer.rs:108-133): everything here is new.

Layers (so the N>1 logic is testable without GPUs):
  * `partition_columns`, `rebalance_boundaries`  — pure numpy
  * `Transport`      — neighbour exchange of byte messages (CUDA tensors on nccl, host
                       tensors on gloo)
  * `SlabDriver`     — the per-step protocol, generic over an engine adapter
  * `HipSlabEngine`  — adapter over the C ABI (`fs_slab_*`); tests inject an oracle adapter
"""
import ctypes as C
import json
import os
import time

import numpy as np

HEADER_BYTES = 16
RECORD_BYTES = 16


# ----------------------------------------------------------------------------- partition
def global_columns(positions_x, bounds_x, h):
    """Global cell column of x (funcs.wgsl:212-214) in f32, for host-side partitioning."""
    f = np.float32
    c = np.floor((positions_x.astype(f) + f(bounds_x) * f(0.5)) / f(h))
    return np.clip(c, 0, 4294967295.0).astype(np.int64) + 1


def partition_columns(hist, world, min_cols=4):
    """Boundaries b[0..world] (b[0]=0, b[world]=len(hist)) giving ~equal particle counts."""
    gw = len(hist)
    csum = np.concatenate([[0], np.cumsum(hist.astype(np.int64))])
    total = csum[-1]
    b = [0]
    for r in range(1, world):
        target = total * r // world
        c = int(np.searchsorted(csum, target, side="left"))
        c = max(c, b[-1] + min_cols)
        c = min(c, gw - (world - r) * min_cols)
        b.append(c)
    b.append(gw)
    return b


def rebalance_boundaries(bounds, hist, max_shift, min_cols=4):
    """Move each interior boundary towards the equal-count partition by at most max_shift columns."""
    world = len(bounds) - 1
    ideal = partition_columns(hist, world, min_cols)
    new = list(bounds)
    for k in range(1, world):
        d = int(np.clip(ideal[k] - bounds[k], -max_shift, max_shift))
        new[k] = bounds[k] + d
    for k in range(1, world):                      # keep slabs at least min_cols wide
        new[k] = max(new[k], new[k - 1] + min_cols)
    for k in range(world - 1, 0, -1):
        new[k] = min(new[k], new[k + 1] - min_cols)
    return new


def trim_outer_edges(bounds, hist, margin, min_cols=4):
    """The first and last slab own everything out to the domain walls, which in a dam break is mostly empty
    space: their local grids (and with them the cell-start table and the counting sort's scan) would be many
    times larger than an inner slab's, and the slowest rank sets the pace.  Pull the two OUTER edges in to
    `margin` columns beyond the occupied columns (never past the walls, never inside a neighbour).  A particle
    that outruns the margin between two re-balancing steps is counted as `lost` by the slab engine, like any
    other protocol violation.  margin <= 0 keeps the walls."""
    gw = len(hist)
    new = list(bounds)
    occ = np.nonzero(np.asarray(hist))[0]
    if margin <= 0 or occ.size == 0:
        new[0], new[-1] = 0, gw
        return new
    new[0] = int(min(max(0, int(occ[0]) - margin), new[1] - min_cols))
    new[-1] = int(max(min(gw, int(occ[-1]) + 1 + margin), new[-2] + min_cols))
    return new


def default_trim_margin():
    """Floor of the outer-edge margin in columns (FS_SLAB_TRIM_MARGIN); the margin actually used is
    max(this, travel_margin(...)) so that it is tied to how far the fluid can move before the next re-trim."""
    return int(os.environ.get("FS_SLAB_TRIM_MARGIN", "256"))


SPEED_CLAMP = 500.0          # compute.wgsl:118-122: |v| <= 500 after every step


def boundary_columns(vmax, accel, dt, h, steps, safety=1.5, floor=4):
    """Width of the boundary zone of an overlapped slab step (fs_slab_set_boundary_cols): a migrant must land at least 3
    columns short of the interior, so 3 + the columns the fastest particle can cross in ONE step at any time before the next
    re-balancing step (`steps` steps away; vmax = largest speed now, all ranks).  Never below `floor`."""
    v_end = min(SPEED_CLAMP, safety * float(vmax) + float(accel) * float(dt) * steps)
    return max(int(floor), 3 + int(np.ceil(v_end * float(dt) / float(h))))


def travel_margin(vmax, accel, dt, h, steps, safety=1.5):
    """Columns a particle can cross in `steps` steps: it moves at most (safety * vmax + accel * dt * k) * dt in
    step k (vmax = largest speed now, all ranks; accel = |gravity|, the one body force; `safety` covers pressure
    pushes), never faster than the engine's speed clamp.  +2: the slab's own 2-column halo tolerance."""
    v_end = min(SPEED_CLAMP, safety * float(vmax) + float(accel) * float(dt) * steps)
    v_mean = min(SPEED_CLAMP, 0.5 * (safety * float(vmax) + v_end))
    return int(np.ceil(steps * v_mean * float(dt) / float(h))) + 2


class SlabProtocolError(RuntimeError):
    """A device counter (lost / overflow / far_halo) is non-zero: particles were dropped or a message / the slot
    array overflowed.  The run is no longer a valid simulation of the scene."""


# ----------------------------------------------------------------------------- transport
class Transport:
    """Exchange one byte message with each slab neighbour (rank-1 = left, rank+1 = right)."""

    def __init__(self, rank, world, message_bytes, device=None, loopback=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = rank, world
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        if loopback:     # tests: both neighbours are this rank itself — what it sends right arrives as its right-hand incoming
            self.left = self.right = rank          # message, likewise left (sends and receives to one peer match in order)
        self.device = device            # torch.device("cuda", i) for nccl, None for host (gloo)
        kw = {"dtype": torch.uint8, "device": device} if device is not None else {"dtype": torch.uint8}
        self.send_left = torch.zeros(message_bytes, **kw)
        self.send_right = torch.zeros(message_bytes, **kw)
        self.recv_left = torch.zeros(message_bytes, **kw)
        self.recv_right = torch.zeros(message_bytes, **kw)
        # the four descriptors never change (fixed tensors, fixed peers): build them once
        self.ops = []
        if self.right is not None:
            self.ops.append(dist.P2POp(dist.isend, self.send_right, self.right))
            self.ops.append(dist.P2POp(dist.irecv, self.recv_right, self.right))
        if self.left is not None:
            self.ops.append(dist.P2POp(dist.isend, self.send_left, self.left))
            self.ops.append(dist.P2POp(dist.irecv, self.recv_left, self.left))

    def exchange(self):
        if not self.ops:
            return
        for req in self.dist.batch_isend_irecv(self.ops):
            req.wait()          # nccl: makes the current stream wait; gloo: blocks the host


class _DevMsg:
    """A device message buffer with the two accessors the engine adapter needs."""

    def __init__(self, g, name, nbytes, device_index):
        self.buf = g.ResizableBuffer(name, np.uint8, nbytes, device=device_index)

    def data_ptr(self):
        return self.buf.device_ptr


class NativeTransport:
    """The same neighbour exchange through the C ABI's own RCCL binding (fs_comm_init / fs_slab_exchange,
    csrc/comm.hip): grouped ncclSend/ncclRecv on the simulation's stream, no torch tensor or torch stream on the
    data path — what a Rust (or any non-Python) host calls.  torch.distributed is used ONCE, to ship rank 0's
    128-byte RCCL id to the other ranks; any rendezvous would do.  Opt-in: FS_NATIVE_RCCL=1 for bench.py --gpus N."""

    def __init__(self, g, rank, world, message_bytes, device_index, dist=None, loopback=False):
        self.g, self.rank, self.world = g, rank, world
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        if loopback:     # tests: fs_slab_exchange with left_rank == right_rank == own rank (recv_right <- send_right, recv_left <- send_left)
            self.left = self.right = rank
        self.device = ("native", device_index)      # not None: the engine adapter writes the messages in place
        self.torch_device = None                    # where the (torch) re-balancing all-reduces run; set by the caller
        self.send_left, self.send_right, self.recv_left, self.recv_right = (
            _DevMsg(g, k, message_bytes, device_index) for k in ("send_left", "send_right", "recv_left", "recv_right"))
        lib = g.load_library()
        idbuf = (C.c_uint8 * 128)()
        if rank == 0:
            g._check(lib, lib.fs_comm_unique_id(idbuf))
        if world > 1:
            box = [bytes(idbuf)]
            dist.broadcast_object_list(box, src=0)
            idbuf = (C.c_uint8 * 128).from_buffer_copy(box[0])
        self.comm = C.c_void_p()
        g._check(lib, lib.fs_comm_init(int(device_index), rank, world, idbuf, C.byref(self.comm)))
        self.lib, self.sim_handle = lib, None
        import torch
        import torch.distributed as tdist
        self.torch, self.dist = torch, tdist        # re-balancing all-reduces stay on torch (tiny, every K steps)

    def bind(self, sim):
        self.sim_handle = sim._h

    def exchange(self):
        P = lambda m: C.c_void_p(m.data_ptr())
        self.g._check(self.lib, self.lib.fs_slab_exchange(
            self.sim_handle, self.comm, -1 if self.left is None else self.left, -1 if self.right is None else self.right,
            P(self.send_left), P(self.send_right), P(self.recv_left), P(self.recv_right)))

    def close(self):
        if self.comm.value:
            self.lib.fs_comm_destroy(self.comm)
            self.comm = C.c_void_p()


# ----------------------------------------------------------------------------- engines
class HipSlabEngine:
    """Adapter over the C ABI.  With a CUDA transport the messages are torch CUDA tensors and
    the kernels write/read them in place (the sim's stream is made torch's current stream so
    RCCL orders after the pack and before the unpack).  With a host transport (gloo) the
    messages are staged through fs_buffer device buffers."""

    def __init__(self, g, settings, bounds, rank, world, capacity, recv_capacity, max_cols, device_index, transport, **sim_kw):
        self.g = g
        self.sim = g.SlabSimulation(settings, bounds[rank], bounds[rank + 1], transport.left is not None, transport.right is not None,
                                    capacity, recv_capacity, max_cols, device=device_index, **sim_kw)
        self.t = transport
        self.cuda = transport.device is not None
        if hasattr(transport, "bind"):
            transport.bind(self.sim)
        if not self.cuda:
            mb = self.sim.message_bytes
            self.dev = {k: g.ResizableBuffer(k, np.uint8, mb, device=device_index)
                        for k in ("send_left", "send_right", "recv_left", "recv_right")}

    @property
    def message_bytes(self):
        return self.sim.message_bytes

    def _ptr(self, name):
        if self.cuda:
            return C.c_void_p(getattr(self.t, name).data_ptr())
        return C.c_void_p(self.dev[name].device_ptr)

    def pack(self, tick):
        self.sim.pack(tick, self._ptr("send_left"), self._ptr("send_right"))
        if not self.cuda:       # stage device -> host tensors for gloo
            self.sim.wait_packed()      # the messages only: an overlapped handle goes on computing its interior columns
            for k in ("send_left", "send_right"):
                getattr(self.t, k).numpy()[:] = self.dev[k].read()

    def exchange(self):
        """The neighbour exchange of this step.  Overlapped handle + device transport: issued on the handle's exchange stream
        (behind the pack, beside the interior columns' kernels); fs_slab_step waits for it."""
        t = self.t
        if not self.sim.overlapped or not self.cuda or isinstance(t, NativeTransport):
            t.exchange()        # host transport / serial handle / fs_slab_exchange (which does the stream hand-over itself)
            return
        if not t.ops:
            return
        if getattr(self, "_comm_ext", None) is None:
            self._comm_ext = t.torch.cuda.ExternalStream(self.sim.comm_stream_ptr, device=t.device)
        self.sim.comm_begin()
        with t.torch.cuda.stream(self._comm_ext):
            t.exchange()
        self.sim.comm_end()

    def set_boundary_cols(self, cols):
        if self.sim.overlapped:
            self.sim.set_boundary_cols(cols)

    def finish(self):
        if not self.cuda:
            for k in ("recv_left", "recv_right"):
                self.dev[k].write(0, getattr(self.t, k).numpy())
        self.sim.step(self._ptr("recv_left"), self._ptr("recv_right"))

    def set_window(self, lo, hi):
        self.sim.set_window(lo, hi)

    def column_histogram(self, gw):
        return self.sim.column_histogram(gw)

    def owned_particles(self):
        rec, owned = self.sim.download()
        return rec[owned]

    def counters(self):
        return self.sim.counters()

    def max_speed(self):
        return self.sim.max_speed()

    def rebalance_inputs(self, gw):
        """(column histogram summed over ranks as int64[gw], [lost, overflow, far_halo, vmax] maxed over ranks) with ONE
        device read: fs_slab_rebalance_stats leaves both on the device, the two all-reduces run on the simulation's
        stream (fs_comm_allreduce with the native transport, torch.distributed on the same stream with nccl), and the
        host reads the reduced buffer once.  Host transport (gloo): one device read, then the all-reduces on the host."""
        g, t = self.g, self.t
        torch_cuda = self.cuda and not isinstance(t, NativeTransport)
        if getattr(self, "_reb", None) is None or self._reb_gw != gw:
            self._reb_gw = gw
            if torch_cuda:
                self._reb = t.torch.zeros(gw + 4, dtype=t.torch.int32, device=t.device)
            else:
                self._reb = g.ResizableBuffer("rebalance", np.uint32, gw + 4, device=self.sim.device_index)
        base = self._reb.data_ptr() if torch_cuda else self._reb.device_ptr
        self.sim.rebalance_stats(C.c_void_p(base + 4 * gw), C.c_void_p(base), gw)
        if torch_cuda:
            self.sim.sync()                        # produced on the simulation's stream, which need not be torch's current one
            t.dist.all_reduce(self._reb[:gw], op=t.dist.ReduceOp.SUM)
            t.dist.all_reduce(self._reb[gw:], op=t.dist.ReduceOp.MAX)
            host = self._reb.cpu().numpy().view(np.uint32)
        elif isinstance(t, NativeTransport):
            lib = t.lib
            g._check(lib, lib.fs_comm_allreduce(self.sim._h, t.comm, C.c_void_p(base), gw, 0, 0))              # u32, SUM
            g._check(lib, lib.fs_comm_allreduce(self.sim._h, t.comm, C.c_void_p(base + 4 * gw), 4, 0, 1))       # u32, MAX
            self.sim.sync()                        # the ONE host synchronisation of a re-balancing step
            host = self._reb.read()
        else:
            self.sim.sync()
            host = self._reb.read()
            th, ts = t.torch.from_numpy(host[:gw].astype(np.int64)), t.torch.from_numpy(host[gw:].astype(np.int64))
            t.dist.all_reduce(th, op=t.dist.ReduceOp.SUM)
            t.dist.all_reduce(ts, op=t.dist.ReduceOp.MAX)
            host = np.concatenate([th.numpy(), ts.numpy()]).astype(np.uint32)
        stats = np.array([host[gw], host[gw + 1], host[gw + 2],
                          float(np.array([host[gw + 3]], dtype=np.uint32).view(np.float32)[0])], dtype=np.float64)
        return host[:gw].astype(np.int64), stats

    def sync(self):
        self.sim.sync()


class SlabDriver:
    """Per-step protocol: pack -> neighbour exchange -> finish; optional re-balancing.

    Every re-balancing step (already a host synchronisation) also (1) reads the device violation counters of
    every rank and raises SlabProtocolError if any is non-zero — a long run can no longer delete particles
    silently; (2) sizes the outer-edge margin from the largest particle speed of all ranks and the number of
    steps until the next re-trim (`travel_margin`), never below `trim_margin`; (3) shortens the interval to the
    next re-balancing step while the partition is badly unbalanced (boundaries move at most `max_shift` columns
    per step, so a rank could otherwise outgrow its slot capacity between two of them)."""

    def __init__(self, engine, transport, bounds, grid_w, rebalance_every=0, max_shift=2, trim_margin=None,
                 capacity_main=None, check_counters=True, max_cols=None):
        self.e, self.t = engine, transport
        self.bounds = list(bounds)                 # bounds[0] / bounds[-1] are the (possibly trimmed) outer edges
        self.grid_w = grid_w
        self.rebalance_every, self.max_shift = rebalance_every, max_shift
        self.trim_margin = default_trim_margin() if trim_margin is None else trim_margin
        self.capacity_main = capacity_main         # main slots per rank (None: no capacity-driven interval)
        self.check_counters = check_counters
        self.max_cols = max_cols                   # widest window the engine was created for (None: the whole grid)
        self.steps = 0
        self.next_rebalance = rebalance_every
        self.last_margin = self.trim_margin
        self.tick = None
        self.initial_vmax = 0.0                    # largest speed of the initial state (callers that start with moving particles set it)
        self.last_boundary = None

    def _set_boundary(self, vmax, interval):
        """Overlapped engines: size the boundary zone for the speeds expected until the next re-balancing step."""
        if not hasattr(self.e, "set_boundary_cols") or self.tick is None:
            return
        g = self.tick.gravity
        cols = boundary_columns(vmax, float(np.hypot(g.x, g.y)), self.tick.delta, self._h(), interval)
        self.last_boundary = cols
        self.e.set_boundary_cols(cols)

    def step(self, tick):
        self.tick = tick
        if self.steps == 0:
            self._set_boundary(self.initial_vmax, self.rebalance_every or 64)
        self.e.pack(tick)           # overlapped engine: the pack AND everything that needs no incoming message
        (getattr(self.e, "exchange", None) or self.t.exchange)()
        self.e.finish()             # overlapped engine: the boundary strips
        self.steps += 1
        if self.rebalance_every and self.steps >= self.next_rebalance:
            self.rebalance()

    def _allreduce(self, arr, op):
        torch, dist = self.t.torch, self.t.dist
        th = torch.from_numpy(arr)
        tdev = getattr(self.t, "torch_device", self.t.device)
        if tdev is not None:
            th = th.to(tdev)
        dist.all_reduce(th, op=op)
        return th.cpu().numpy()

    def rebalance(self):
        dist = self.t.dist
        c = {}
        # device-resident engine: two all-reduces, ONE host read.  FS_REBALANCE_HOST=1 keeps the older path (three blocking
        # reads + two all-reduces of host arrays); the torch-nccl and the fs_comm_allreduce branches of rebalance_inputs have
        # only ever run single-rank (no multi-GPU box): tests/test_multi_gpu.py compares the two paths over gloo, world 2
        if hasattr(self.e, "rebalance_inputs") and not os.environ.get("FS_REBALANCE_HOST"):
            hist, stats = self.e.rebalance_inputs(self.grid_w)
            if not self.check_counters:
                stats[:3] = 0
            c = {"max over ranks": [int(x) for x in stats[:3]]}
        else:
            hist = self.e.column_histogram(self.grid_w).astype(np.int64)
            hist = self._allreduce(hist, dist.ReduceOp.SUM)          # tiny (grid_w * 8 B), every K steps only
            # violations + largest speed of any rank, one more tiny all-reduce (MAX)
            c = self.e.counters() if (self.check_counters and hasattr(self.e, "counters")) else {}
            vmax = float(self.e.max_speed()) if hasattr(self.e, "max_speed") else SPEED_CLAMP
            stats = np.array([c.get("lost", 0), c.get("overflow", 0), c.get("far_halo", 0), vmax], dtype=np.float64)
            stats = self._allreduce(stats, dist.ReduceOp.MAX)
        if stats[:3].any():
            raise SlabProtocolError(f"slab protocol violated at step {self.steps}: max over ranks of lost/overflow/far_halo = "
                                    f"{int(stats[0])}/{int(stats[1])}/{int(stats[2])} (this rank: {c})")
        world = len(self.bounds) - 1
        owned = np.array([hist[self.bounds[k]:self.bounds[k + 1]].sum() for k in range(world)], dtype=np.float64)
        interval = self.rebalance_every
        if owned.sum() > 0:
            worst = owned.max()
            limit = 0.9 * self.capacity_main if self.capacity_main else 1.15 * owned.mean()
            if worst > limit:                     # catching up: boundaries move <= max_shift columns per re-balancing step
                interval = max(1, self.rebalance_every // 8)
            elif worst > 1.10 * owned.mean():
                interval = max(1, self.rebalance_every // 2)
        self.next_rebalance = self.steps + interval
        new = rebalance_boundaries(self.bounds, hist, self.max_shift)
        margin = self.trim_margin
        if margin > 0 and self.tick is not None:
            g = self.tick.gravity
            accel = float(np.hypot(g.x, g.y))
            margin = max(margin, travel_margin(stats[3], accel, self.tick.delta, self._h(), interval))
        self.last_margin = margin
        self._set_boundary(stats[3], interval)
        new = trim_outer_edges(new, hist, margin)               # outer edges follow the occupied columns
        if self.max_cols:                                       # never wider than the engine's tables
            new[0] = max(new[0], new[1] - self.max_cols)
            new[-1] = min(new[-1], new[-2] + self.max_cols)
        if new != self.bounds:
            self.bounds = new
            self.e.set_window(new[self.t.rank], new[self.t.rank + 1])

    def _h(self):
        st = getattr(self.e, "settings", None) or getattr(getattr(self.e, "sim", None), "settings", None)
        return float(st.smoothing_radius) if st is not None else 1.0


# ----------------------------------------------------------------------------- setup helpers
def slab_capacities(n_total, world, grid_h, headroom=1.25):
    """(capacity, recv_capacity).  A message carries 2 ghost columns + migrants: ~2 * grid_h * 4 records at
    lattice density; 3 * grid_h * 8 leaves ~3x headroom (the device counters flag an overflow).  Smaller
    messages matter: they are sent at full size every step (no host sync to learn the real count).
    Main slots hold the worst-case owned share (mean * headroom — SlabDriver re-balances more often once a rank
    passes 90 % of it) plus the 4 ghost columns that are live after a step (2 per side, 10 particles per cell)."""
    recv = max(4096, 3 * grid_h * 8)
    main = int(n_total / world * headroom) + 4 * grid_h * 10 + 4096
    return main + 2 * recv, recv


def main_slots(capacity, recv_capacity):
    return capacity - 2 * recv_capacity


def initial_owned(g, settings, offset, bounds, rank):
    """This rank's share of the reference lattice (src/simulation.rs:147-163) + column histogram."""
    lat = g.reference_lattice(settings, offset)
    cols = global_columns(lat["position"][:, 0], settings.size.x, settings.smoothing_radius)
    lo, hi = bounds[rank], bounds[rank + 1]
    return lat[(cols >= lo) & (cols < hi)]


def lattice_histogram(g, settings, offset):
    lat = g.reference_lattice(settings, offset)
    cols = global_columns(lat["position"][:, 0], settings.size.x, settings.smoothing_radius)
    gw = int(np.ceil(np.float32(settings.size.x) / np.float32(settings.smoothing_radius))) + 2
    return np.bincount(cols, minlength=gw)[:gw].astype(np.int64), gw


def scaling_base(g, settings, off, tick, gh, gw, device_index, warmup, steps):
    """ms/step of the slab engine as ONE rank over the whole domain (no neighbours, no exchange), same window."""
    import ctypes as C
    n = settings.particle_count
    cap, recv = slab_capacities(n, 1, gh)
    sim = g.SlabSimulation(settings, 0, gw, False, False, cap, recv, gw, device=device_index)
    sim.upload_owned(g.reference_lattice(settings, off))
    for _ in range(warmup):
        sim.pack(tick, None, None); sim.step(None, None)
    sim.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.pack(tick, None, None); sim.step(None, None)
    sim.sync()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    ok = sim.counters()
    sim.close()
    return {"ms_per_step": round(ms, 4), "value": round(n / (ms * 1e-3) / 1e6, 2), "unit": "M particle-steps/s",
            "n_gpus": 1, "engine": "slab engine, 1 rank, counting sort", "violations": ok["lost"] + ok["overflow"] + ok["far_halo"]}


# ----------------------------------------------------------------------------- bench entry
def bench_main(args, rank, local_rank, world):
    """bench.py --gpus N (N > 1): strong scaling of the 16M dam break over N slabs."""
    import torch
    import torch.distributed as dist
    import gpu_fluid_simulation_amd as g
    from bench import ALG_BYTES, ALG_TOTAL, HBM_COPY_GBS, HBM_PEAK_GBS, WORKLOADS, bound_from_evidence, load_json

    backend = os.environ.get("FS_DIST_BACKEND", "nccl")
    if os.environ.get("FS_FORCE_DEVICE0"):      # single-GPU rehearsal: every rank on device 0 (gloo only)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world,
                            device_id=torch.device("cuda", local_rank) if backend == "nccl" else None)
    n = WORKLOADS[args.workload]
    settings, off, tick = g.dam_break_2d(n)
    hist, gw = lattice_histogram(g, settings, off)
    gh = int(np.ceil(np.float32(settings.size.y) / np.float32(settings.smoothing_radius))) + 2
    bounds = partition_columns(hist, world)
    cap, recv = slab_capacities(n, world, gh)
    max_cols = gw      # tables for the whole grid (42 MB each at 16M): the outer-edge margin follows the fluid's speed
    rebalance_every = int(os.environ.get("FS_REBALANCE_EVERY", "64"))
    # outer slabs: occupied columns + margin — only while re-balancing keeps moving the edges with the fluid
    bounds = trim_outer_edges(bounds, hist, default_trim_margin() if rebalance_every > 0 else 0)
    msg_bytes = HEADER_BYTES + RECORD_BYTES * recv
    dev = torch.device("cuda", local_rank) if backend == "nccl" else None
    native = bool(os.environ.get("FS_NATIVE_RCCL")) and backend == "nccl"
    if native:      # the C ABI's own RCCL binding (fs_slab_exchange): what a non-Python host would run
        tr = NativeTransport(g, rank, world, msg_bytes, local_rank, dist)
        tr.torch_device = dev
    else:
        tr = Transport(rank, world, msg_bytes, device=dev)
    eng = HipSlabEngine(g, settings, bounds, rank, world, cap, recv, max_cols, local_rank, tr)
    assert eng.message_bytes == msg_bytes
    eng.sim.upload_owned(initial_owned(g, settings, off, bounds, rank))
    drv = SlabDriver(eng, tr, bounds, gw, rebalance_every=rebalance_every, capacity_main=main_slots(cap, recv), max_cols=max_cols)

    ext = torch.cuda.ExternalStream(eng.sim.stream_ptr, device=torch.device("cuda", local_rank))
    with torch.cuda.stream(ext):
        for _ in range(args.warmup):
            drv.step(tick)
        eng.sync()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            drv.step(tick)
        eng.sync()
        torch.cuda.synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        # after the timed region: a short window with a HIP event at every pass boundary of THIS rank's stream (all ranks
        # keep stepping in lockstep), for the roofline of the dominant pass
        eng.sim.profile(True)
        eng.sim.profile_read(reset=True)
        prof_steps = max(1, min(args.steps, 20))
        for _ in range(prof_steps):
            drv.step(tick)
        eng.sync()
        passes, psteps = eng.sim.profile_read(reset=True)
        eng.sim.profile(False)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if dev is not None else "cpu")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    cnt = eng.counters()
    bad = torch.tensor([cnt["lost"] + cnt["overflow"] + cnt["far_halo"]], dtype=torch.int64,
                       device=dev if dev is not None else "cpu")
    dist.all_reduce(bad)
    nlive = torch.tensor([int(eng.owned_particles().shape[0])], dtype=torch.int64,
                         device=dev if dev is not None else "cpu")
    dist.all_reduce(nlive)
    base = None
    if rank == 0 and not os.environ.get("FS_NO_SCALING_BASE"):
        base = scaling_base(g, settings, off, tick, gh, gw, local_rank, args.warmup, args.steps)
    if rank == 0:
        ms_per_step = float(tmax.item()) * 1e3 / args.steps
        value = n / (ms_per_step * 1e-3) / 1e6
        agg = ALG_TOTAL * n / (ms_per_step * 1e-3) / 1e9
        # dominant pass of rank 0 (its share of the particles): algorithmic bytes / its measured time; the bound comes
        # from the committed counter summary of that pass's kernel, as for N = 1 — never assumed
        n_rank = int(cnt["n_live"])
        alg = dict(ALG_BYTES, predict_key=28, sort=12)           # slabs: pack (predict + key + messages) is its own pass
        pp = {k: v / max(psteps, 1) for k, v in passes.items() if k in alg}
        # the `predict_key` interval of a slab step is pack + the halo exchange + unpack: communication, not a kernel —
        # it is reported beside the roofline (pack_exchange_ms); the roofline's kernel is the longest COMPUTE pass
        kernels_only = {k: v for k, v in pp.items() if k != "predict_key"} or pp
        dom = max(kernels_only, key=kernels_only.get)
        dom_gbs = alg[dom] * n_rank / (pp[dom] * 1e-3) / 1e9 if pp[dom] > 0 else 0.0
        bound, crow = bound_from_evidence(dom, dom_gbs / HBM_COPY_GBS, load_json("counters_latest.json"), False)
        out = {
            "metric": "M particle-steps/s", "value": round(value, 2), "unit": "M particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "particles": n, "scene": "SURVEY.md §8d dam_break_2d",
                       "sort": "counting (slab default: per-rank sorts are tolerance-parity by construction)",
                       "parallelism": f"{world} column slabs, RCCL p2p halo ({backend}" + (", C-ABI fs_slab_exchange)" if native else ", torch.distributed)"),
                       "slab_columns": [drv.bounds[k + 1] - drv.bounds[k] for k in range(world)],
                       "outer_trim_margin": drv.trim_margin,
                       "message_bytes": msg_bytes},
            "roofline": {"bound": bound, "kernel": f"{dom} pass of rank 0 ({n_rank} live particles incl. ghosts)",
                         "achieved": round(dom_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(dom_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                         "valu_issue_frac_of_kernel": crow.get("valu_issue_frac") if crow else None,
                         "rank0_passes_ms": {k: round(v, 4) for k, v in pp.items()},
                         "pack_exchange_ms": round(pp.get("predict_key", 0.0), 4),
                         "exchange_longer_than_kernel": bool(pp.get("predict_key", 0.0) > pp[dom]),
                         "step_aggregate": {"alg_bytes_per_particle": ALG_TOTAL, "achieved": round(agg, 1),
                                            "peak": HBM_PEAK_GBS * world, "frac": round(agg / (HBM_PEAK_GBS * world), 4)}},
            "checks": {"particles_conserved": int(nlive.item()) == n, "protocol_violations": int(bad.item())},
            "slab_step": {0: "serial (pack -> exchange -> step)", 1: "edge-first (next step's messages built and exchanged beside the interior "
                          "columns' force pass; column-major cell ids on ranks with neighbours)", 2: "strips (interior while the messages "
                          "fly, boundary strips afterwards)"}[eng.sim.step_mode],
            "boundary_cols": drv.last_boundary,
            # like-for-like base of the scaling curve: N = 1 of bench.py runs the PLAIN engine with the reference
            # network; this is the SAME slab engine (counting sort, fixed-capacity slots, pack/unpack) as ONE rank
            "scaling_base": base,
            "multi_gpu_note": "8-GPU timing is produced by the driver's SCALE run only; this builder's boxes have one GPU",
        }
        if not args.no_cpu_baseline:     # rank 0, after the timed region (the other ranks wait in the barrier below)
            from bench import cpu_baseline
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
