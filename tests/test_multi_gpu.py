"""HIP slab engine (fs_slab_*) on the GPU: several slabs of one scene, all on device 0,
exchanging their device messages directly, compared with the single-GPU engine (tolerance
parity: SURVEY §8e) — plus a 2-process gloo rehearsal of bench.py's multi-rank path."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class InProcessSlabs:
    def __init__(self, fs, settings, off, world, cap, recv, seed=None, vel=1.0, sort_mode=None, trim_margin=0, serial=False,
                 boundary_cols=None, strips=False):
        from gpu_fluid_simulation_amd import multi
        self.fs, self.multi, self.world, self.trim_margin = fs, multi, world, trim_margin
        lat = fs.reference_lattice(settings, off)
        if seed is not None:
            rng = np.random.default_rng(seed)
            lat["position"] += rng.uniform(-0.025, 0.025, size=lat["position"].shape).astype(np.float32)
            lat["predicted_position"] = lat["position"]
            lat["velocity"] = rng.uniform(-vel, vel, size=lat["velocity"].shape).astype(np.float32)
        self.initial = lat
        cols = multi.global_columns(lat["position"][:, 0], settings.size.x, settings.smoothing_radius)
        self.gw = int(np.ceil(np.float32(settings.size.x) / np.float32(settings.smoothing_radius))) + 2
        hist = np.bincount(cols, minlength=self.gw)[: self.gw]
        self.bounds = multi.trim_outer_edges(multi.partition_columns(hist, world), hist, trim_margin)
        self.sims, self.bufs = [], []
        for r in range(world):
            s = fs.SlabSimulation(settings, self.bounds[r], self.bounds[r + 1], r > 0, r < world - 1, cap, recv,
                                  max_cols=self.gw, device=0, sort_mode=sort_mode, serial=serial, strips=strips)
            assert s.step_mode == (0 if serial or sort_mode == fs.FS_SORT_BITONIC else 2 if strips else 1)
            if boundary_cols is not None and s.overlapped:
                s.set_boundary_cols(boundary_cols)
            s.upload_owned(lat[(cols >= self.bounds[r]) & (cols < self.bounds[r + 1])])
            self.sims.append(s)
            self.bufs.append({k: fs.ResizableBuffer(k, np.uint8, s.message_bytes) for k in ("sl", "sr", "rl", "rr")})

    def step(self, tick):
        P = lambda b: C.c_void_p(b.device_ptr)
        for r, s in enumerate(self.sims):
            s.pack(tick, P(self.bufs[r]["sl"]), P(self.bufs[r]["sr"]))
        for s in self.sims:
            s.wait_packed()                           # messages complete before they are moved
        # the exchange: every rank's outgoing messages are COPIED into its neighbours' incoming buffers (an edge-first step
        # refills its outgoing buffers with the next tick's messages before it returns, so they cannot be shared)
        for r in range(self.world):
            if r > 0:
                self.bufs[r]["rl"].write(0, self.bufs[r - 1]["sr"].read())
            if r < self.world - 1:
                self.bufs[r]["rr"].write(0, self.bufs[r + 1]["sl"].read())
        for r, s in enumerate(self.sims):
            s.step(P(self.bufs[r]["rl"]) if r > 0 else None, P(self.bufs[r]["rr"]) if r < self.world - 1 else None)
        for s in self.sims:
            s.sync()

    def rebalance(self, max_shift, tick=None, interval=None):
        """As multi.SlabDriver.rebalance: histogram, violation counters, outer-edge margin from the largest speed."""
        hist = np.zeros(self.gw, dtype=np.int64)
        for s in self.sims:
            hist += s.column_histogram(self.gw)
        self.assert_clean()
        new = self.multi.rebalance_boundaries(self.bounds, hist, max_shift)
        margin = self.trim_margin
        if margin > 0 and tick is not None and interval:
            vmax = max(s.max_speed() for s in self.sims)
            accel = float(np.hypot(tick.gravity.x, tick.gravity.y))
            margin = max(margin, self.multi.travel_margin(vmax, accel, tick.delta, self.sims[0].settings.smoothing_radius, interval))
        self.last_margin = margin
        new = self.multi.trim_outer_edges(new, hist, margin)
        for r, s in enumerate(self.sims):
            s.set_window(new[r], new[r + 1])
        self.bounds = new
        return hist

    def owned(self):
        out = []
        for s in self.sims:
            rec, own = s.download()
            out.append(rec[own])
        return np.concatenate(out)

    def assert_clean(self):
        for s in self.sims:
            c = s.counters()
            assert c["lost"] == 0 and c["overflow"] == 0 and c["far_halo"] == 0, c


@pytest.mark.parametrize("mode", ["edge", "strips", "serial"])
@pytest.mark.parametrize("world,n,seed", [(2, 4096, None), (3, 4096, 7), (4, 16384, 3), (8, 65536, 5)])
def test_slabs_match_single_gpu(fs, world, n, seed, mode):
    """edge (default): edge columns first, next step's messages built and exchanged beside the interior columns' force pass;
    strips: interior while the messages fly, boundary strips afterwards; serial: the round-3 step."""
    from tests.slab_oracle import assert_statistics_close, match_and_compare
    st, off, tick = fs.dam_break_2d(n)
    slabs = InProcessSlabs(fs, st, off, world, cap=n + 4 * 2048, recv=2048, seed=seed, serial=mode == "serial", strips=mode == "strips")
    single = fs.FluidSimulation(st, device=0, initial_offset=off, ref_quirks=False)
    single.upload_particles(slabs.initial)
    assert slabs.owned().shape[0] == n
    for s in range(24):
        slabs.step(tick)
        single.tick(tick)
        if s in (0, 1, 4):
            slabs.assert_clean()
            # steps 1-2: cell keys of matched particles IDENTICAL (north_star: cell indices bit-exact); step 5:
            # particles that sit exactly on a cell boundary may have flipped with 1-ulp x differences
            match_and_compare(slabs.owned(), single.download_particles(), st.smoothing_radius,
                              max_key_flips=0.0 if s < 2 else 0.02)
    slabs.assert_clean()
    assert_statistics_close(slabs.owned(), single.download_particles(), n)


def test_trimmed_outer_edges_follow_the_fluid(fs):
    """Outer slabs own only the occupied columns + a margin (multi.trim_outer_edges); the edges are moved out
    again at every re-balancing step while the dam break runs across the domain: nothing is lost, parity holds."""
    from tests.slab_oracle import assert_statistics_close, match_and_compare
    n = 16384
    st, off, tick = fs.dam_break_2d(n)
    slabs = InProcessSlabs(fs, st, off, 4, cap=n + 4 * 2048, recv=2048, trim_margin=6)
    gw = slabs.gw
    assert slabs.bounds[-1] < gw, "the far edge must start well inside the (mostly empty) domain"
    single = fs.FluidSimulation(st, device=0, initial_offset=off, ref_quirks=False)
    single.upload_particles(slabs.initial)
    first_edge = slabs.bounds[-1]
    for s in range(1, 161):
        slabs.step(tick)
        single.tick(tick)
        if s % 4 == 0:                                  # margin 6 columns: re-balance before the front can cross it
            slabs.rebalance(2)
        if s in (2, 4):
            slabs.assert_clean()
            match_and_compare(slabs.owned(), single.download_particles(), st.smoothing_radius,
                              max_key_flips=0.0 if s == 2 else 0.02)
    slabs.assert_clean()
    assert slabs.owned().shape[0] == n
    assert slabs.bounds[-1] > first_edge, "the front has moved, so must the edge"
    # 160 steps: the tolerances the scene's chaos sets (tests/slab_oracle.py assert_statistics_close)
    assert_statistics_close(slabs.owned(), single.download_particles(), n, pos_atol=1e-2, vel_atol=5e-2, vmax_rtol=None)


def test_slot_capacity_overflow_is_counted(fs):
    """owned particles sorted past the main slots (n_live = owned + ghosts > capacity - 2*recv) used to be
    overwritten by the next unpack without any counter moving.  They are now counted in `overflow` at the next pack
    (multi.SlabDriver raises on it), and a capacity sized by multi.slab_capacities never gets there."""
    n = 16384
    st, off, tick = fs.dam_break_2d(n)
    from gpu_fluid_simulation_amd import multi
    gh = int(np.ceil(np.float32(st.size.y) / np.float32(st.smoothing_radius))) + 2
    cap, rc = multi.slab_capacities(n, 2, gh)
    roomy = InProcessSlabs(fs, st, off, 2, cap=cap, recv=rc, serial=True)
    owned_max = max(s.counters()["n_live"] for s in roomy.sims)       # before the first step: the uploaded owned particles
    for _ in range(12):
        roomy.step(tick)
    roomy.assert_clean()
    assert roomy.owned().shape[0] == n
    recv = 2048
    main = owned_max + 64                                  # room for the owned share, none for the ~500 ghosts
    tight = InProcessSlabs(fs, st, off, 2, cap=main + 2 * recv, recv=recv, serial=True)
    tight.step(tick)
    live = [s.counters()["n_live"] for s in tight.sims]
    assert max(live) > main, (live, main)                  # the step left more live records than main slots
    tight.step(tick)                                       # ... which the next pack must notice
    assert sum(s.counters()["overflow"] for s in tight.sims) > 0
    with pytest.raises(fs.FluidSimError):                  # more owned particles than main slots: rejected up front
        tight.sims[0].upload_owned(tight.initial[: main + 1])


def _cell_histograms_equal(a, b, ncell):
    ha, hb = np.bincount(a["grid"], minlength=ncell), np.bincount(b["grid"], minlength=ncell)
    bad = int((ha != hb).sum())
    if bad:
        print(f"{bad} of {ncell} cells differ in occupancy")
    return bad == 0


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_geometry_16m_slabs(fs, world):
    """The slab protocol at bench.py --gpus N's real geometry (16M scene, capacities and message sizes from
    multi.slab_capacities, outer edges trimmed, re-balanced every 64 steps), all slabs on one GPU, over the bench window
    10 + 100 steps: no lost / overflow / far-halo event, 16 777 216 particles conserved, and at steps 1-2 exactly the
    same number of particles in every cell as the single-GPU engine (cell keys bit-exact)."""
    from gpu_fluid_simulation_amd import multi
    n = 1 << 24
    st, off, tick = fs.dam_break_2d(n)
    gh = int(np.ceil(np.float32(st.size.y) / np.float32(st.smoothing_radius))) + 2
    cap, recv = multi.slab_capacities(n, world, gh)
    slabs = InProcessSlabs(fs, st, off, world, cap=cap, recv=recv, trim_margin=multi.default_trim_margin())
    single = fs.FluidSimulation(st, device=0, initial_offset=off, ref_quirks=False, sort_mode=fs.FS_SORT_COUNTING)
    for s in range(1, 111):
        slabs.step(tick)
        if s <= 2:
            single.tick(tick)
            assert _cell_histograms_equal(slabs.owned(), single.download_particles(), slabs.gw * gh)
        if s == 2:
            single.close()
        if s % 64 == 0:
            slabs.rebalance(2, tick=tick, interval=64)
    slabs.assert_clean()
    own = slabs.owned()
    assert own.shape[0] == n
    assert np.isfinite(own["position"]).all() and np.isfinite(own["velocity"]).all()


def test_config4_64m_eight_slabs(fs):
    """BASELINE configs[4] through its own path: the 64M-particle dam break as 8 column slabs (all on this one GPU, the
    messages handed over directly), 12 steps including one re-balancing step: particles conserved, no lost / overflow /
    far-halo event; at steps 1-2 every cell holds exactly as many particles as in the single-GPU run (cell keys
    bit-exact; matching 64M particles pairwise is not needed for that) and density / position statistics agree."""
    from gpu_fluid_simulation_amd import multi
    n, world = 1 << 26, 8
    st, off, tick = fs.dam_break_2d(n)
    gh = int(np.ceil(np.float32(st.size.y) / np.float32(st.smoothing_radius))) + 2
    cap, recv = multi.slab_capacities(n, world, gh)
    slabs = InProcessSlabs(fs, st, off, world, cap=cap, recv=recv, trim_margin=multi.default_trim_margin())
    del slabs.initial
    assert slabs.gw == 8194 and gh == 5122
    single = fs.FluidSimulation(st, device=0, initial_offset=off, ref_quirks=False, sort_mode=fs.FS_SORT_COUNTING)
    for s in range(1, 13):
        slabs.step(tick)
        if s <= 2:
            single.tick(tick)
            a, b = slabs.owned(), single.download_particles()
            assert a.shape[0] == n
            assert _cell_histograms_equal(a, b, slabs.gw * gh), f"cell occupancy differs at step {s}"
            np.testing.assert_allclose(a["density"].mean(dtype=np.float64), b["density"].mean(dtype=np.float64), rtol=1e-6)
            np.testing.assert_allclose(a["position"].mean(axis=0, dtype=np.float64), b["position"].mean(axis=0, dtype=np.float64), atol=1e-6)
            np.testing.assert_allclose(a["velocity"].mean(axis=0, dtype=np.float64), b["velocity"].mean(axis=0, dtype=np.float64), atol=1e-6)
            del a, b
        if s == 2:
            single.close()
        if s == 6:
            before = list(slabs.bounds)
            slabs.rebalance(2, tick=tick, interval=6)
            assert slabs.last_margin >= multi.default_trim_margin()
            assert len(before) == world + 1
    slabs.assert_clean()
    own = slabs.owned()
    assert own.shape[0] == n
    assert np.isfinite(own["position"]).all() and np.isfinite(own["velocity"]).all()
    assert float(own["density"].min()) >= 0.1


def test_download_between_window_change_and_step_is_consistent(fs):
    """fs_slab_set_window only takes effect at the next pack: a download in between must still translate the
    stored (old-window) keys to the same global cell ids."""
    n = 16384
    st, off, tick = fs.dam_break_2d(n)
    slabs = InProcessSlabs(fs, st, off, 3, cap=n + 4 * 2048, recv=2048, trim_margin=6)
    for _ in range(6):
        slabs.step(tick)
    before = [s.download() for s in slabs.sims]
    new = list(slabs.bounds)
    new[0] = max(0, new[0] - 3); new[1] += 1; new[2] -= 1; new[3] += 5      # every window moves or resizes
    for r, sim in enumerate(slabs.sims):
        sim.set_window(new[r], new[r + 1])
    slabs.bounds = new
    after = [s.download() for s in slabs.sims]
    for (ra, oa), (rb, ob) in zip(before, after):
        assert np.array_equal(oa, ob) and np.array_equal(ra["grid"], rb["grid"])
        assert np.array_equal(ra["position"], rb["position"])
    slabs.step(tick)                                    # and the run goes on cleanly in the new windows
    slabs.assert_clean()
    assert slabs.owned().shape[0] == n


def test_single_slab_bitonic_equals_plain_engine(fs):
    """With the reference network selected, one slab over the whole domain is bit-identical to the plain
    engine: DEAD keys sort last and never move, so the live prefix gets the same permutation."""
    n = 16384
    st, off, tick = fs.dam_break_2d(n)
    slabs = InProcessSlabs(fs, st, off, 1, cap=n + 2 * 2048 + 999, recv=2048, sort_mode=fs.FS_SORT_BITONIC)
    single = fs.FluidSimulation(st, device=0, initial_offset=off, ref_quirks=False)
    for _ in range(9):
        slabs.step(tick)
        single.tick(tick)
    assert np.array_equal(slabs.owned().view(np.uint8), single.download_particles().view(np.uint8))


def test_slab_rebalancing_keeps_parity_and_conserves(fs):
    from tests.slab_oracle import assert_statistics_close, match_and_compare
    n = 16384
    st, off, tick = fs.dam_break_2d(n)
    slabs = InProcessSlabs(fs, st, off, 4, cap=n + 4 * 4096, recv=4096, seed=11, vel=3.0)
    single = fs.FluidSimulation(st, device=0, initial_offset=off, ref_quirks=False)
    single.upload_particles(slabs.initial)
    b0 = list(slabs.bounds)
    for s in range(40):
        slabs.step(tick)
        single.tick(tick)
        if s % 2 == 1:
            hist = slabs.rebalance(max_shift=1)
            assert hist.sum() == n                      # column histogram sees every owned particle once
        if s == 1:
            match_and_compare(slabs.owned(), single.download_particles(), st.smoothing_radius, max_key_flips=0.0)
        if s == 5:                                      # boundaries have moved three times by now
            match_and_compare(slabs.owned(), single.download_particles(), st.smoothing_radius, max_key_flips=0.02)
    slabs.assert_clean()
    assert_statistics_close(slabs.owned(), single.download_particles(), n)


@pytest.mark.parametrize("ranks", [2, 3])
def test_multi_rank_bench_rehearsal_gloo(fs, ranks):
    """bench.py --gpus N launched as N ranks that share GPU 0 (RCCL refuses several ranks on one
    device, so the rehearsal uses gloo with host-staged messages; the driver's real run uses nccl).
    3 ranks gives a middle rank with two neighbours; re-balancing runs every 3 steps."""
    env = dict(os.environ, FS_DIST_BACKEND="gloo", OMP_NUM_THREADS="2", FS_REBALANCE_EVERY="3")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(29610 + ranks), "bench.py", "--gpus", str(ranks), "--steps", "8",
           "--warmup", "3", "--workload", "dam_break_2d_1M", "--no-cpu-baseline"]
    env["FS_FORCE_DEVICE0"] = "1"
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == ranks and d["value"] > 0
    assert d["checks"]["particles_conserved"] and d["checks"]["protocol_violations"] == 0


NCCL_SMOKE = r'''
import os, sys, ctypes as C
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist          # torch FIRST: one HIP runtime per process
import numpy as np
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import multi
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
st, off, tick = g.dam_break_2d(16384)
hist, gw = multi.lattice_histogram(g, st, off)
bounds = multi.partition_columns(hist, 1)
dev = torch.device("cuda", 0)
tr = multi.Transport(0, 1, multi.HEADER_BYTES + multi.RECORD_BYTES * 4096, device=dev)
eng = multi.HipSlabEngine(g, st, bounds, 0, 1, 16384 + 4 * 4096, 4096, gw, 0, tr)
eng.sim.upload_owned(multi.initial_owned(g, st, off, bounds, 0))
drv = multi.SlabDriver(eng, tr, bounds, gw, rebalance_every=3)
ext = torch.cuda.ExternalStream(eng.sim.stream_ptr, device=dev)
with torch.cuda.stream(ext):
    for _ in range(7):
        drv.step(tick)
    eng.sync()
    t = torch.ones(4, device=dev)
    dist.all_reduce(t)                            # RCCL on the sim's stream
    torch.cuda.synchronize()
own = eng.owned_particles()
single = g.FluidSimulation(st, device=0, initial_offset=off, ref_quirks=False, sort_mode=g.FS_SORT_COUNTING)
for _ in range(7):
    single.tick(tick)
ref = single.download_particles()
assert own.shape[0] == 16384 and float(t[0]) == 1.0
# one slab covering the whole domain runs the same stable counting sort over the same slot order as the
# plain engine in FS_SORT_COUNTING mode -> bit-exact
assert np.array_equal(own.view(np.uint8), ref.view(np.uint8)), "single slab differs from the plain engine"
print("nccl smoke ok", eng.counters())
dist.destroy_process_group()
'''


def test_nccl_single_rank_smoke(fs, tmp_path):
    script = tmp_path / "nccl_smoke.py"
    script.write_text(NCCL_SMOKE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29633")
    out = subprocess.run([sys.executable, str(script), ROOT], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "nccl smoke ok" in out.stdout


NATIVE_SELF = r'''
import ctypes as C, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import gpu_fluid_simulation_amd as g          # no torch in this process: the C ABI alone drives RCCL
lib = g.load_library()
st, off, tick = g.dam_break_2d(16384)
sim = g.SlabSimulation(st, 10, 40, False, False, 16384 + 2 * 2048, 2048, 66, device=0)
nb = sim.message_bytes
bufs = {k: g.ResizableBuffer(k, np.uint8, nb) for k in ("sl", "sr", "rl", "rr")}
rng = np.random.default_rng(3)
pat = {k: rng.integers(0, 256, nb, dtype=np.uint8) for k in ("sl", "sr")}
bufs["sl"].write(0, pat["sl"]); bufs["sr"].write(0, pat["sr"])
idb = (C.c_uint8 * 128)()
g._check(lib, lib.fs_comm_unique_id(idb))
comm = C.c_void_p()
g._check(lib, lib.fs_comm_init(0, 0, 1, idb, C.byref(comm)))
P = lambda b: C.c_void_p(b.device_ptr)
for _ in range(3):
    g._check(lib, lib.fs_slab_exchange(sim._h, comm, 0, 0, P(bufs["sl"]), P(bufs["sr"]), P(bufs["rl"]), P(bufs["rr"])))
sim.sync()
assert np.array_equal(bufs["rr"].read(), pat["sr"]) and np.array_equal(bufs["rl"].read(), pat["sl"])
# in-place all-reduce on the simulation's stream (single rank: identity) — the histogram path of a native host
h = g.ResizableBuffer("hist", np.uint32, 64)
h.write(0, np.arange(64, dtype=np.uint32))
g._check(lib, lib.fs_comm_allreduce(sim._h, comm, C.c_void_p(h.device_ptr), 64, 0, 0))
sim.sync()
assert np.array_equal(h.read(), np.arange(64, dtype=np.uint32))
# no neighbours: nothing is issued, nothing fails
g._check(lib, lib.fs_slab_exchange(sim._h, comm, -1, -1, None, None, None, None))
lib.fs_comm_destroy(comm)
print("native rccl self-exchange ok", nb)
'''


def test_native_rccl_self_exchange_through_the_c_abi(fs, tmp_path):
    """fs_comm_* / fs_slab_exchange (csrc/comm.hip): grouped ncclSend/ncclRecv on the simulation's stream, driven
    from a process that never imports torch — a single-rank communicator exchanging both messages with itself."""
    script = tmp_path / "native_self.py"
    script.write_text(NATIVE_SELF)
    out = subprocess.run([sys.executable, str(script), ROOT], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "native rccl self-exchange ok" in out.stdout


def _one_slab(fs, n, own_lo_frac, own_hi_frac, recv, seed=3, vel=3.0, **kw):
    """A slab in the middle of the domain with both neighbours present: returns (sim, settings, tick, owned records)."""
    from gpu_fluid_simulation_amd import multi
    st, off, tick = fs.dam_break_2d(n)
    lat = fs.reference_lattice(st, off)
    rng = np.random.default_rng(seed)
    lat["position"] += rng.uniform(-0.04, 0.04, size=lat["position"].shape).astype(np.float32)
    lat["predicted_position"] = lat["position"]
    lat["velocity"] = rng.uniform(-vel, vel, size=lat["velocity"].shape).astype(np.float32)
    cols = multi.global_columns(lat["position"][:, 0], st.size.x, st.smoothing_radius)
    occ = np.nonzero(np.bincount(cols))[0]
    lo = int(occ[0] + own_lo_frac * (occ[-1] - occ[0])); hi = int(occ[0] + own_hi_frac * (occ[-1] - occ[0]))
    own = lat[(cols >= lo) & (cols < hi)]
    gw = int(np.ceil(np.float32(st.size.x) / np.float32(st.smoothing_radius))) + 2
    cap = own.shape[0] + 2 * recv + 4096
    sim = fs.SlabSimulation(st, lo, hi, True, True, cap, recv, max_cols=gw, device=0, **kw)
    sim.upload_owned(own)
    return sim, st, tick, own, (lo, hi)


def _read_message(fs, buf, message_bytes):
    raw = buf.read()
    hdr = raw[:16].view(np.uint32)
    rec = raw[16:].view(np.float32).reshape(-1, 4)
    return int(hdr[0]), int(hdr[1]), rec


def test_pack_messages_are_in_slot_order_and_deterministic(fs):
    """k_slab_pack + k_slab_msg: the records each neighbour gets are the flagged particles in SLOT order (the look-back
    offsets make the compaction deterministic), identical from run to run, and exactly the particles whose predicted
    column lies in the 2-column band at the slab edge."""
    from gpu_fluid_simulation_amd import multi
    outs = []
    for _ in range(2):
        sim, st, tick, own, (lo, hi) = _one_slab(fs, 65536, 0.30, 0.55, recv=8192)
        sl = fs.ResizableBuffer("sl", np.uint8, sim.message_bytes); sr = fs.ResizableBuffer("sr", np.uint8, sim.message_bytes)
        sim.pack(tick, C.c_void_p(sl.device_ptr), C.c_void_p(sr.device_ptr)); sim.sync()
        cl, ol, rl = _read_message(fs, sl, sim.message_bytes)
        cr, orr, rr = _read_message(fs, sr, sim.message_bytes)
        assert ol == 0 and orr == 0 and cl > 0 and cr > 0
        outs.append((rl[:cl].copy(), rr[:cr].copy()))
        # expected: the uploaded (slot-ordered) records whose predicted column is within 2 of the edge
        dt = np.float32(tick.delta)
        pred_x = own["position"][:, 0] + own["velocity"][:, 0] * dt
        bs = np.float32(st.size.x) * np.float32(0.5)
        pred_x = np.where(np.abs(pred_x) > bs, bs * np.sign(pred_x), pred_x).astype(np.float32)
        pc = multi.global_columns(pred_x, st.size.x, st.smoothing_radius)
        want_l = own[pc < lo + 2]; want_r = own[pc + 2 >= hi]
        assert cl == want_l.shape[0] and cr == want_r.shape[0]
        assert np.array_equal(rl[:cl, :2].view(np.uint32), want_l["position"].view(np.uint32))
        assert np.array_equal(rr[:cr, 2:].view(np.uint32), want_r["velocity"].view(np.uint32))
        sim.close()
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))


def test_message_overflow_is_flagged_in_header_and_counter(fs):
    """More records than a message holds: count == capacity, header overflow flag set, fs_slab_counters.overflow != 0."""
    sim, st, tick, own, _ = _one_slab(fs, 65536, 0.30, 0.55, recv=64)
    sl = fs.ResizableBuffer("sl", np.uint8, sim.message_bytes); sr = fs.ResizableBuffer("sr", np.uint8, sim.message_bytes)
    sim.pack(tick, C.c_void_p(sl.device_ptr), C.c_void_p(sr.device_ptr)); sim.sync()
    cl, ol, _ = _read_message(fs, sl, sim.message_bytes)
    cr, orr, _ = _read_message(fs, sr, sim.message_bytes)
    assert (cl, ol) == (64, 1) and (cr, orr) == (64, 1)
    empty = fs.ResizableBuffer("e", np.uint8, sim.message_bytes)      # zero header: no records
    sim.step(C.c_void_p(empty.device_ptr), C.c_void_p(empty.device_ptr)); sim.sync()
    assert sim.counters()["overflow"] != 0
    sim.close()


def test_rebalance_stats_on_device_equal_the_blocking_reads(fs):
    """fs_slab_rebalance_stats leaves {lost, overflow, far_halo, max-speed bits} and the column histogram in device
    buffers: same numbers as fs_slab_counters_read / fs_slab_max_speed / fs_slab_column_histogram."""
    sim, st, tick, own, (lo, hi) = _one_slab(fs, 65536, 0.30, 0.55, recv=8192)
    empty = fs.ResizableBuffer("e", np.uint8, sim.message_bytes)
    sl = fs.ResizableBuffer("sl", np.uint8, sim.message_bytes); sr = fs.ResizableBuffer("sr", np.uint8, sim.message_bytes)
    for _ in range(3):
        sim.pack(tick, C.c_void_p(sl.device_ptr), C.c_void_p(sr.device_ptr))
        sim.step(C.c_void_p(empty.device_ptr), C.c_void_p(empty.device_ptr))
    sim.sync()
    gw = int(np.ceil(np.float32(st.size.x) / np.float32(st.smoothing_radius))) + 2
    buf = fs.ResizableBuffer("reb", np.uint32, gw + 4)
    sim.rebalance_stats(C.c_void_p(buf.device_ptr + 4 * gw), C.c_void_p(buf.device_ptr), gw)
    sim.sync()                                     # enqueued on the simulation's stream; the buffer read is not
    got = buf.read()
    c = sim.counters()
    assert list(got[gw:gw + 3]) == [c["lost"], c["overflow"], c["far_halo"]]
    assert got[gw + 3:gw + 4].view(np.float32)[0] == np.float32(sim.max_speed())
    assert np.array_equal(got[:gw], sim.column_histogram(gw))
    assert got[:gw].sum() > 0 and got[:lo].sum() == 0 and got[hi:gw].sum() == 0
    sim.close()


def test_edge_first_prebuilt_messages_equal_a_full_pack(fs):
    """Edge-first step (the default): the messages of tick t + 1 are built at the end of step t from the edge columns the first
    force launch advanced, on the exchange stream, beside the interior columns' force launch.  They must be byte for byte what a
    full classification of every slot (the serial step's fs_slab_pack) produces from the same state — and the state itself must
    not depend on how the force pass was split."""
    sims, bufs = [], []
    for serial in (False, True):
        sim, st, tick, own, (lo, hi) = _one_slab(fs, 65536, 0.30, 0.55, recv=8192, serial=serial)
        assert sim.step_mode == (0 if serial else 1)
        sims.append(sim)
        bufs.append({k: fs.ResizableBuffer(k, np.uint8, sim.message_bytes) for k in ("sl", "sr", "e")})
    P = lambda b: C.c_void_p(b.device_ptr)
    for step in range(6):
        msgs = []
        for sim, b in zip(sims, bufs):
            sim.pack(tick, P(b["sl"]), P(b["sr"]))       # edge-first, step >= 1: finds the pre-built messages, classifies for the sort only
            sim.wait_packed()
            msgs.append((b["sl"].read(), b["sr"].read()))
        for (el, er), (fl, fr) in [(msgs[0], msgs[1])]:
            for e, f in ((el, fl), (er, fr)):
                cnt = int(e[:16].view(np.uint32)[0])
                assert np.array_equal(e[:16], f[:16]) and cnt > 0, (step, e[:16].view(np.uint32), f[:16].view(np.uint32))
                assert np.array_equal(e[16:16 + 16 * cnt], f[16:16 + 16 * cnt]), f"step {step}: pre-built message differs from the full pack"
        for sim, b in zip(sims, bufs):
            sim.step(P(b["e"]), P(b["e"]))               # empty incoming messages (zero header)
            sim.sync()
        a, oa = sims[0].download(); c, oc = sims[1].download()
        assert np.array_equal(oa, oc) and np.array_equal(a[oa].view(np.uint8), c[oc].view(np.uint8)), f"step {step}: states differ"
    for sim in sims:
        c = sim.counters()
        assert c["far_halo"] == 0 and c["overflow"] == 0
        sim.close()


def test_edge_first_state_replaced_or_window_moved_between_steps(fs):
    """What an edge-first step leaves behind for the next one — pre-built messages, the edge columns' slots already classified
    (their histogram counts included) — is void when the state is replaced (fs_slab_upload_owned) or the window moves
    (fs_slab_set_window) in between: the next pack must then do the whole job, and end up byte for byte where the serial step does."""
    sims, bufs = [], []
    for serial in (False, True):
        sim, st, tick, own, (lo, hi) = _one_slab(fs, 65536, 0.30, 0.55, recv=8192, serial=serial)
        sims.append(sim)
        bufs.append({k: fs.ResizableBuffer(k, np.uint8, sim.message_bytes) for k in ("sl", "sr", "e")})
    P = lambda b: C.c_void_p(b.device_ptr)

    def cycle(label):
        msgs = []
        for sim, b in zip(sims, bufs):
            sim.pack(tick, P(b["sl"]), P(b["sr"]))
            sim.wait_packed()
            msgs.append((b["sl"].read(), b["sr"].read()))
        for e, f in zip(msgs[0], msgs[1]):
            cnt = int(f[:16].view(np.uint32)[0])
            assert np.array_equal(e[:16 + 16 * cnt], f[:16 + 16 * cnt]), f"{label}: messages differ"
        for sim, b in zip(sims, bufs):
            sim.step(P(b["e"]), P(b["e"]))
            sim.sync()
        a, oa = sims[0].download(); c, oc = sims[1].download()
        assert np.array_equal(oa, oc) and np.array_equal(a[oa].view(np.uint8), c[oc].view(np.uint8)), f"{label}: states differ"
        return a[oa]

    for k in range(3):
        cycle(f"warm-up {k}")
    # 1. the state is replaced: the same particles shifted by a third of a cell, velocities reversed
    rep = own.copy()
    rep["position"][:, 0] += np.float32(0.07)
    rep["predicted_position"] = rep["position"]
    rep["velocity"] *= np.float32(-1.0)
    for sim in sims:
        sim.upload_owned(rep)
    for k in range(3):
        cycle(f"after the upload {k}")
    # 2. the window moves by one column on either side (what a re-balancing does)
    for sim in sims:
        sim.set_window(lo + 1, hi - 1)
    for k in range(3):
        last = cycle(f"after the window move {k}")
    assert last.shape[0] > 0
    for sim in sims:
        c = sim.counters()
        assert c["overflow"] == 0, c
        sim.close()


def test_edge_zone_narrower_than_the_travel_is_counted(fs):
    """A particle that reaches the 2-column halo band from farther inside than the edge zone is missing from the pre-built message;
    the next fs_slab_pack — which classifies every slot anyway — counts it in far_halo (multi.SlabDriver raises on it), and a zone
    sized by multi.boundary_columns for that speed stays clean."""
    from gpu_fluid_simulation_amd import multi
    speed = 60.0                                         # 2.5 columns per step at h = 0.2, dt = 1/120
    for cols, clean in ((3, False), (multi.boundary_columns(speed, 0.0, 1 / 120, 0.2, 1), True)):
        sim, st, tick, own, (lo, hi) = _one_slab(fs, 65536, 0.30, 0.55, recv=16384, vel=0.0)
        tick.gravity = fs.Vec2(0.0, 0.0)
        own = own.copy()
        own["velocity"][:, 0] = -speed
        sim.upload_owned(own)
        sim.set_boundary_cols(cols)
        b = {k: fs.ResizableBuffer(k, np.uint8, sim.message_bytes) for k in ("sl", "sr", "e")}
        P = lambda x: C.c_void_p(x.device_ptr)
        for _ in range(3):
            sim.pack(tick, P(b["sl"]), P(b["sr"]))
            sim.step(P(b["e"]), P(b["e"]))
        sim.pack(tick, P(b["sl"]), P(b["sr"])); sim.step(P(b["e"]), P(b["e"])); sim.sync()
        c = sim.counters()
        assert (c["far_halo"] == 0) == clean, (cols, c)
        sim.close()


REBALANCE_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import multi
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 65536
st, off, tick = g.dam_break_2d(n)
hist, gw = multi.lattice_histogram(g, st, off)
gh = int(np.ceil(np.float32(st.size.y) / np.float32(st.smoothing_radius))) + 2
bounds = multi.partition_columns(hist, world)
cap, recv = multi.slab_capacities(n, world, gh)
tr = multi.Transport(rank, world, multi.HEADER_BYTES + multi.RECORD_BYTES * recv)
eng = multi.HipSlabEngine(g, st, bounds, rank, world, cap, recv, gw, 0, tr)
eng.sim.upload_owned(multi.initial_owned(g, st, off, bounds, rank))
drv = multi.SlabDriver(eng, tr, bounds, gw, rebalance_every=0)
for _ in range(7):
    drv.step(tick)
eng.sync()
# the device-resident path: fs_slab_rebalance_stats + two all-reduces + ONE read ...
h_dev, s_dev = eng.rebalance_inputs(gw)
# ... against the host path it replaced: three blocking reads, two all-reduces of host arrays
h_host = torch.from_numpy(eng.column_histogram(gw).astype(np.int64)); dist.all_reduce(h_host)
c = eng.counters()
s_host = torch.tensor([c["lost"], c["overflow"], c["far_halo"], float(eng.max_speed())], dtype=torch.float64)
dist.all_reduce(s_host, op=dist.ReduceOp.MAX)
assert np.array_equal(h_dev, h_host.numpy()), "column histograms differ"
assert int(h_dev.sum()) == n, int(h_dev.sum())
assert np.array_equal(s_dev[:3], s_host.numpy()[:3]) and np.float32(s_dev[3]) == np.float32(s_host.numpy()[3]), (s_dev, s_host)
assert s_dev[3] > 0
# and the driver's two ways of re-balancing move the boundaries identically
b0 = list(drv.bounds)
drv.rebalance_every = 1; drv.rebalance(); via_dev = list(drv.bounds)
dist.barrier()
print("rebalance paths agree", rank, b0, via_dev, flush=True)
dist.destroy_process_group()
"""


def test_rebalance_inputs_equal_the_host_path_world2(fs, tmp_path):
    """HipSlabEngine.rebalance_inputs (device buffers, all-reduce, one read) against the column_histogram + counters +
    max_speed + host all-reduce path it replaced, with TWO ranks (gloo, both on this GPU): identical histogram (summing to every
    particle), identical violation counters and largest speed.  Its torch-nccl and fs_comm_allreduce branches need N GPUs and
    stay unverified on hardware."""
    script = tmp_path / "reb_worker.py"
    script.write_text(REBALANCE_WORKER)
    import socket
    with socket.socket() as sk:          # a port nobody holds right now (a fixed one can sit in TIME_WAIT from an earlier run)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), cwd=ROOT, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert all("rebalance paths agree" in o for o in outs)


TORCH_VS_NATIVE = r"""
import os, sys, ctypes as C
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist          # torch FIRST: one HIP runtime per process
import numpy as np
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import multi
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
n = 65536
st, off, tick = g.dam_break_2d(n)
hist, gw = multi.lattice_histogram(g, st, off)
occ = np.nonzero(hist)[0]
lo, hi = int(occ[0] + 0.3 * (occ[-1] - occ[0])), int(occ[0] + 0.55 * (occ[-1] - occ[0]))
bounds = [lo, hi]
lat = g.reference_lattice(st, off)
rng = np.random.default_rng(4)
lat["velocity"] = rng.uniform(-3, 3, size=lat["velocity"].shape).astype(np.float32)
cols = multi.global_columns(lat["position"][:, 0], st.size.x, st.smoothing_radius)
own = lat[(cols >= lo) & (cols < hi)]
recv = 8192
cap = own.shape[0] + 2 * recv + 4096
mb = multi.HEADER_BYTES + multi.RECORD_BYTES * recv
states = {}
for kind in ("torch", "native"):
    tr = (multi.Transport(0, 1, mb, device=dev, loopback=True) if kind == "torch"
          else multi.NativeTransport(g, 0, 1, mb, 0, dist, loopback=True))
    eng = multi.HipSlabEngine(g, st, bounds, 0, 1, cap, recv, gw, 0, tr)
    assert eng.sim.step_mode == 1 and eng.sim.cfg.has_left and eng.sim.cfg.has_right
    eng.sim.upload_owned(own)
    drv = multi.SlabDriver(eng, tr, [lo, hi], gw, rebalance_every=0, check_counters=False)
    ext = torch.cuda.ExternalStream(eng.sim.stream_ptr, device=dev)
    with torch.cuda.stream(ext):
        for _ in range(8):
            drv.step(tick)          # no synchronisation in between: the exchange of step t + 1 runs beside step t's interior launch
        eng.sync()
        torch.cuda.synchronize()
    rec, owned = eng.sim.download()
    states[kind] = (rec.copy(), owned.copy(), eng.counters())
    if kind == "native":
        tr.close()
a, b = states["torch"], states["native"]
assert a[2] == b[2], (a[2], b[2])
assert np.array_equal(a[1], b[1]) and np.array_equal(a[0][a[1]].view(np.uint8), b[0][b[1]].view(np.uint8)), "torch-nccl and fs_slab_exchange runs differ"
assert a[1].sum() > 1000
print("torch nccl exchange on the handle's exchange stream == fs_slab_exchange", a[2])
dist.destroy_process_group()
"""


def test_torch_nccl_exchange_on_the_exchange_stream_equals_the_native_one(fs, tmp_path):
    """bench.py --gpus N moves the messages with torch.distributed (nccl = RCCL) issued on the handle's exchange stream between
    fs_slab_comm_begin / _end (multi.HipSlabEngine.exchange).  One GPU cannot host two nccl ranks, so the plumbing is exercised with
    both neighbours being the rank itself (not a valid simulation: its own edge particles come back as migrants) and compared
    with the same loop through fs_slab_exchange: eight free-running edge-first steps, byte-equal states and counters."""
    script = tmp_path / "torch_vs_native.py"
    script.write_text(TORCH_VS_NATIVE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29644")
    out = subprocess.run([sys.executable, str(script), ROOT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "== fs_slab_exchange" in out.stdout
