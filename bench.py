#!/usr/bin/env python3
"""bench.py — headline benchmark: M particle-steps/s on the 16M-particle 2D dam break
(BASELINE.json `metric`, configs[2]; SURVEY.md §8d scene), one process per GPU.

A "step" is one pass of the hot path (predict -> key -> bitonic sort -> cell starts ->
density -> force+integrate) over all particles.  State is resident in HBM before the
timed region; timing uses HIP events on the simulation's own stream (C ABI
fs_timed_steps / fs_profile_*), bracketed by barrier + device sync, MAX over ranks.

Prints ONE JSON line on rank 0 (contract fields + `roofline` + `cpu_baseline`), plus — never as
the headline — `alt_modes` (opt-in engine modes), `alt_windows` (the same engine over later windows
of the same scene, where the fluid is disordered / dense) and `alt_workloads` (the other
single-GPU configs of BASELINE.json).

Profiling: `rocprofv3 ... -- python3 bench.py --no-build ...` — never build inside a profiled
process (the profiler's preload has initialised the GPU; a build would exec compilers from it).
Build first with `python __graft_entry__.py`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md §8d — algorithmic (compulsory SoA) bytes per particle-step, by pass.  predict + key
# (28 B) run inside the first sort kernel, so their bytes are the sort pass's: 28 + 12 = 40.
ALG_BYTES = {"sort": 28 + 12, "reorder": 48 + 4, "density": 16, "force": 48}
ALG_TOTAL = 156                      # 28 + 60 + 4 + 16 + 48
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0

WORKLOADS = {
    "dam_break_2d_16M": 1 << 24,
    "dam_break_2d_1M": 1 << 20,
    "dam_break_2d_4096": 4096,
    "dam_break_2d_64M": 1 << 26,
    "dam_break_3d_8M": 200 ** 3,
}
ALG_BYTES_3D = {"sort": 40 + 12, "reorder": 72 + 4, "density": 20, "force": 68}   # 216 B; predict + key (40 B) run inside the first sort kernel, as in 2D
ALG_TOTAL_3D = 216

# which kernel carries a pass: EXACT names as rocprofv3 prints them (profiles/counters_latest.json keys), first match wins
PASS_KERNEL = {"force": ["fsd::k_force<0, false>"], "density": ["fsd::k_density<false, true>", "fsd::k_density<false, false>", "fsd::k_density<false>"],
               "sort": ["fsd::k_bitonic_local32<1, 4>", "fsd::k_bitonic_local32<1, 3>"], "reorder": ["fsd::k_reorder<true>"]}
PASS_KERNEL_3D = {"force": ["fsd::k3_force<0>"], "density": ["fsd::k3_density<0>"],
                  "sort": ["fsd::k_bitonic_local32<2, 4>"], "reorder": ["fsd::k3_reorder"]}
PASS_KERNEL_3D_TOL = {"force": ["fsd::k3_force<2>"], "density": ["fsd::k3_density<2>"],
                      "sort": ["fsd::k_bitonic_local32<2, 4>"], "reorder": ["fsd::k3_reorder"]}


def usable_cores():
    """CPU share of this process: cgroup quota if there is one (a GPU box gives each job a slice of a
    256-thread host), else the affinity mask; capped at 16 as the box's per-GPU share."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(seconds_budget=15.0):
    """The CPU oracle (C++ port of the reference step) timed on this host on a bounded sample of the same
    scene (1M-particle dam break, as many steps as fit): the scalar port (1 thread) is THE cpu_baseline;
    the same port with OpenMP over particles on all cores is reported beside it."""
    import gpu_fluid_simulation_amd as g
    from oracle import oracle as O
    O.build()                           # the checker is built by its users, not by the product build
    n = 1 << 20
    st, off, tick = g.dam_break_2d(n)

    def run(threads, budget):
        O.set_threads(threads)
        sim = O.OracleSim(st, off)
        sim.step(tick)                  # warm-up (page faults, first sort of the lattice)
        steps, t0 = 0, time.perf_counter()
        while True:
            sim.step(tick)
            steps += 1
            el = time.perf_counter() - t0
            if el > budget or steps >= 64:
                break
        sim.close()
        O.set_threads(1)
        return n * steps / el / 1e6, steps, el

    v1, s1, e1 = run(1, seconds_budget)
    cores = min(usable_cores(), O.max_threads())
    vN, sN, eN = run(cores, seconds_budget / 3)
    return {"value": round(v1, 4), "unit": "M particle-steps/s", "cores": 1, "kind": "port",
            "sample": f"dam_break_2d 1M particles, {s1} steps after 1 warm-up, oracle/sph_oracle.cpp scalar, "
                      f"{e1:.1f} s on {os.cpu_count()} host cores (1 used)",
            "all_cores": {"value": round(vN, 3), "cores": cores,
                          "sample": f"same port, OpenMP over particles, {sN} steps in {eN:.1f} s"}}


def under_profiler():
    pre = os.environ.get("LD_PRELOAD", "")
    return ("rocprofiler" in pre or "rocprof" in pre or any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ))


def load_product(no_build):
    """Build (default) or only load the HIP extension.  Under a profiler, or with --no-build, nothing is ever
    compiled: a missing or stale library is an error (exit 3), not a fallback."""
    import __graft_entry__ as ge
    if no_build or under_profiler():
        if ge.product_is_stale() and not os.environ.get("FS_ALLOW_STALE"):
            sys.stderr.write("bench.py --no-build: gpu-fluid-simulation_amd/libfluidsim_hip.so is missing or older than its "
                             "sources; run `python __graft_entry__.py` first (FS_ALLOW_STALE=1 overrides)\n")
            raise SystemExit(3)
        import gpu_fluid_simulation_amd as g
        g.load_library()
        return g
    ge.build_product()
    import gpu_fluid_simulation_amd as g
    return g


def per_pass_table(passes, steps, n, alg_bytes):
    out = {}
    for name, tot in passes.items():
        if name not in alg_bytes:       # FS_PASS_PREDICT_KEY: fused into the sort's first kernel, an empty interval
            continue
        t = tot / steps
        gbs = alg_bytes[name] * n / (t * 1e-3) / 1e9 if t > 0 else 0.0
        out[name] = {"ms": round(t, 4), "alg_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    return out


def run_window(make_sim, tick, warmup, steps, n, alg_bytes, profiled=True):
    """`warmup` untimed steps, then `steps` timed ones on a fresh handle.  Returns (ms_per_step, per-pass table)."""
    sim = make_sim()
    for _ in range(warmup):
        sim.tick(tick)
    sim.sync()
    table = None
    if profiled:
        sim.profile(True)
        sim.profile_read(reset=True)
    ms = sim.timed_steps(tick, steps)
    sim.sync()
    if profiled:
        passes, psteps = sim.profile_read(reset=True)
        assert psteps == steps
        table = per_pass_table(passes, steps, n, alg_bytes)
        sim.profile(False)
    sim.close()
    return ms / steps, table


def load_json(name):
    p = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(p))
    except (OSError, ValueError):
        return None


def bound_from_evidence(dom, hbm_frac_of_copy, counters, is3d, names=None):
    """What limits the dominant kernel, from the committed rocprofv3 counter summary (profiles/counters_latest.json,
    produced by tools/pmc_counters.py): HBM when the algorithmic rate is near the measured copy rate, else the VALU
    issue slots when they are mostly busy, else latency (waves parked on memory / LDS)."""
    if counters is None:
        return "hbm" if hbm_frac_of_copy >= 0.6 else "unknown (no counter summary committed)", None
    row = None
    for name in (names or (PASS_KERNEL_3D if is3d else PASS_KERNEL)).get(dom, [dom]):     # exact kernel names only (no substring match)
        v = counters.get("kernels", {}).get(name)
        if v is not None:
            row = dict(v, kernel=name)
            break
    if row is None:
        return "hbm" if hbm_frac_of_copy >= 0.6 else "unknown (kernel not in the counter summary)", None
    if hbm_frac_of_copy >= 0.6:
        b = "hbm"
    elif row.get("valu_issue_frac", 0.0) >= 0.6:
        b = "valu"
    elif row.get("lds_busy_frac", 0.0) >= 0.6:
        b = "lds"
    else:
        b = "latency"
    return b, row


def spawn_ranks(n, argv, child=None, env_extra=None, timeout=None):
    """`python3 bench.py --gpus N` without a launcher: start N rank processes ourselves (what
    `python -m torch.distributed.run --nproc-per-node N` would do) and relay rank 0's JSON line.

    Runs in a parent that has NOT imported torch or touched HIP, and never replaces itself (no exec): every
    rank is a fresh child with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set.  Rank 0's stdout
    is passed through; the other ranks' stdout goes to stderr.  Returns the exit code: 0 only if every rank
    exited 0; when one rank fails the others are terminated (by PID) instead of waiting in a collective."""
    import socket
    import subprocess
    with socket.socket() as sk:                     # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = child if child is not None else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FS_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this pool
        env.update(env_extra or {})
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=None if r == 0 else sys.stderr))
    t_end = None if timeout is None else time.monotonic() + timeout
    rc, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write(f"bench.py: rank {r} exited with {code}; stopping the other ranks\n")
        timed_out = t_end is not None and time.monotonic() > t_end
        if (rc != 0 or timed_out) and live:
            if timed_out and rc == 0:
                rc = 124
                sys.stderr.write("bench.py: ranks timed out\n")
            for r in live:
                procs[r].terminate()
            for r in live:
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        if live:
            time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # >= 10 warm-up + >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="dam_break_2d_16M", choices=sorted(WORKLOADS))
    ap.add_argument("--sort", default="bitonic", choices=["bitonic", "counting"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip alt_modes / alt_windows / alt_workloads")
    ap.add_argument("--no-build", action="store_true",
                    help="never compile: load the prebuilt library or exit 3 (use under rocprofv3)")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc run")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python3 bench.py --gpus N`: this process becomes the launcher.  Nothing here imports torch, loads the
        # HIP library or initialises the GPU; the build (hipcc child processes) happens before any rank exists.
        if under_profiler():
            # the profiler's preloaded library has initialised the GPU in THIS process: starting ranks from it would be the
            # fork + exec the pool forbids, and none of them would be the profiled program
            sys.stderr.write("bench.py --gpus N under a profiler: profile one rank directly (set RANK / LOCAL_RANK / WORLD_SIZE / "
                             "MASTER_ADDR / MASTER_PORT and put `python3 bench.py ...` after `--`), not the launcher\n")
            raise SystemExit(4)
        if not args.no_build:
            import __graft_entry__ as ge
            ge._load_build_module().build(verbose=False)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if world != args.gpus and not (world > 1 and args.gpus == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU "
                         "(torch.distributed.run), or run bare `python3 bench.py --gpus N`")
    if world > 1:
        # torch first: its bundled HIP runtime has the same SONAME as /opt/rocm's, so the engine
        # library binds to the one already loaded (one runtime per process).
        import torch  # noqa: F401
    g = load_product(args.no_build)

    if world > 1:
        from gpu_fluid_simulation_amd import multi
        return multi.bench_main(args, rank, local_rank, world)

    n = WORKLOADS[args.workload]
    is3d = args.workload.startswith("dam_break_3d")
    alg_bytes, alg_total = (ALG_BYTES_3D, ALG_TOTAL_3D) if is3d else (ALG_BYTES, ALG_TOTAL)
    if is3d:
        st, off, tick = g.dam_break_3d(n)
        make_sim = lambda: g.FluidSimulation3D(st, device=local_rank, initial_offset=off)
    else:
        st, off, tick = g.dam_break_2d(n)
        sort_mode = g.FS_SORT_BITONIC if args.sort == "bitonic" else g.FS_SORT_COUNTING
        make_sim = lambda: g.FluidSimulation(st, device=local_rank, initial_offset=off, sort_mode=sort_mode)

    # headline window: W warm-up steps, then EXACTLY K steps between two HIP events on the sim's stream
    sim = make_sim()
    for _ in range(args.warmup):
        sim.tick(tick)
    sim.sync()                                      # device idle: all work lives on the sim's stream
    t_wall = time.perf_counter()
    ms = sim.timed_steps(tick, args.steps)
    sim.sync()
    t_wall = (time.perf_counter() - t_wall) * 1e3
    # which late-stage plan the sorts of this handle took (csrc/kernels_sort.hip; the result does not depend on it)
    sort_plan = sim.sort_plan() if (not is3d and args.sort == "bitonic") else None
    sim.close()
    ms_per_step = ms / args.steps
    value = n / (ms_per_step * 1e-3) / 1e6            # M particle-steps/s

    # roofline: the SAME window once more on a fresh handle, now with a HIP event at every pass boundary
    # (they serialise the kernel boundaries and cost ~1 %, which is why the headline window runs without them)
    ms_profiled, per_pass = run_window(make_sim, tick, args.warmup, args.steps, n, alg_bytes)
    # the dominant pass = the longest event interval of THIS run (force = the lean kernel + the general one; the per-kernel
    # split of the same command is the committed rocprofv3 summary, profiles/*_kernel_stats.csv)
    dom = max(per_pass, key=lambda k: per_pass[k]["ms"])

    traffic, traffic_src = args.pmc_traffic, "--pmc-traffic argument (separate rocprofv3 --pmc run of this command)"
    if traffic is None:
        traffic_src = None
        tl = load_json("traffic_latest.json")
        if tl and not is3d and tl.get("particles") == n:
            # HBM bytes per step of the dominant pass from the committed rocprofv3 --pmc runs (separate
            # FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction) — NOT measured in this run
            traffic = tl["bytes_per_step_by_pass"].get(dom)
            traffic_src = "profiles/traffic_latest.json (committed; " + tl.get("window", "10+10-step window") + "), not this run"
    counters = load_json("counters_latest.json")
    bound, crow = bound_from_evidence(dom, per_pass[dom]["alg_GBps"] / HBM_COPY_GBS, counters, is3d)
    roofline = {
        "bound": bound, "kernel": dom + (" (k_force + k_force_general)" if dom == "force" and not is3d else ""),
        "achieved": per_pass[dom]["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "profiled_ms_per_step": round(ms_profiled, 4),
        "frac": per_pass[dom]["frac"],
        "traffic": traffic, "traffic_source": traffic_src,
        "alg_bytes_per_particle": alg_bytes[dom],
        "step": {"alg_bytes_per_particle": alg_total,
                 "achieved": round(alg_total * n / (ms_per_step * 1e-3) / 1e9, 1),
                 "frac": round(alg_total * n / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "frac_of_measured_copy": round(alg_total * n / (ms_per_step * 1e-3) / 1e9 / HBM_COPY_GBS, 4)},
        "passes": per_pass,
    }
    if crow is not None:
        # second fraction: share of the VALU issue slots the dominant kernel uses (SQ_INSTS_VALU priced with the
        # measured issue table, / (1024 SIMDs x kernel cycles)), from the committed counter summary named in `source`
        roofline["valu_issue"] = {"frac": crow.get("valu_issue_frac"), "insts_per_wave": crow.get("valu_insts_per_wave"),
                                  "waves_per_simd": crow.get("waves_per_simd"), "lds_busy_frac": crow.get("lds_busy_frac"),
                                  "wait_frac": crow.get("wait_frac"), "kernel": crow.get("kernel"),
                                  "source": crow.get("source") or counters.get("source")}
    out = {
        "metric": "M particle-steps/s", "value": round(value, 2), "unit": "M particle-steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "particles": n, "scene": "SURVEY.md §8d " + ("dam_break_3d (no reference counterpart)" if is3d else "dam_break_2d"),
                   "sort": args.sort, "ref_quirks": not is3d, "parallelism": "1 GPU",
                   "window": f"steps {args.warmup}..{args.warmup + args.steps} from the lattice (see alt_windows for later regimes)"},
        "host_wall_ms_per_step": round(t_wall / args.steps, 4),
        "roofline": roofline,
    }
    if sort_plan is not None:
        out["config"]["sort_plan"] = dict(sort_plan, note="sorts of the warm-up + timed steps by late-stage plan: 'shifted' = one "
                                          "shifted merge replaced stages >= 'stage' (device-side certificate held), 'per_stage' = it "
                                          "failed and the per-stage passes ran; bit-identical arrangement either way")
    if not is3d and args.sort == "bitonic" and not args.no_alt:
        # extras (NOT the headline): the same scene and protocol in the engine's opt-in modes
        def alt_run(**kw):
            mk = lambda: g.FluidSimulation(st, device=local_rank, initial_offset=off, **kw)
            t, _ = run_window(mk, tick, args.warmup, args.steps, n, alg_bytes, profiled=False)
            return {"value": round(n / (t * 1e-3) / 1e6, 2), "unit": "M particle-steps/s", "ms_per_step": round(t, 4)}
        out["alt_modes"] = {
            "counting_sort": dict(alt_run(sort_mode=g.FS_SORT_COUNTING),
                                  note="stable O(N) cell sort instead of the reference network (SURVEY 8f-1); floats equal "
                                       "the headline path to summation-order tolerance"),
            "wgsl_ulp_math": dict(alt_run(math_mode=g.FS_MATH_WGSL_ULP),
                                  note="native rcp/sqrt in the force pass (<= ~1.5 ulp, inside WGSL's 2.5-ULP division "
                                       "contract for the reference shaders); not bit-exact vs the IEEE oracle"),
            "counting_sort+wgsl_ulp_math": alt_run(sort_mode=g.FS_SORT_COUNTING, math_mode=g.FS_MATH_WGSL_ULP),
            "tolerance_math": dict(alt_run(math_mode=g.FS_MATH_TOLERANCE),
                                   note="FS_MATH_TOLERANCE: density / force terms re-associated (FMA, one rsqrt per pair, pressure and "
                                        "1/rho precomputed): within rtol 1e-5 / atol 1e-4*h of the IEEE oracle per step, cell keys and "
                                        "start_indices bit-exact (north_star's float contract); reference sort"),
            "counting_sort+tolerance_math": alt_run(sort_mode=g.FS_SORT_COUNTING, math_mode=g.FS_MATH_TOLERANCE),
        }
        # the same strict engine over later windows of the same scene: the block stays a near lattice for the
        # first ~25 steps; by step 100 it is disordered, from step ~150 the bottom of the column is dense
        aw = {}
        for w0, w1 in ((10, 110), (150, 250)):
            t, tab = run_window(make_sim, tick, w0, w1 - w0, n, alg_bytes)
            aw[f"steps_{w0}_{w1}"] = {"value": round(n / (t * 1e-3) / 1e6, 2), "ms_per_step": round(t, 4),
                                      "passes_ms": {k: v["ms"] for k, v in tab.items()},
                                      "step_frac_of_hbm_peak": round(alg_total * n / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        out["alt_windows"] = dict(aw, note="per-pass events on (costs ~1 %); value in M particle-steps/s")
        # the other single-GPU configs of BASELINE.json, 10 + 100 steps each
        wl = {}
        for name in ("dam_break_2d_1M", "dam_break_2d_64M", "dam_break_3d_8M", "dam_break_3d_8M+tolerance_math"):
            if name == args.workload:
                continue
            try:
                m = WORKLOADS[name.split("+")[0]]
                if name.startswith("dam_break_3d"):
                    s3, o3, t3 = g.dam_break_3d(m)
                    mm3 = g.FS_MATH_TOLERANCE if name.endswith("tolerance_math") else g.FS_MATH_IEEE
                    mk, tk, ab, at = (lambda: g.FluidSimulation3D(s3, device=local_rank, initial_offset=o3, math_mode=mm3)), t3, ALG_BYTES_3D, ALG_TOTAL_3D
                else:
                    s2, o2, t2 = g.dam_break_2d(m)
                    mk, tk, ab, at = (lambda: g.FluidSimulation(s2, device=local_rank, initial_offset=o2)), t2, ALG_BYTES, ALG_TOTAL
                t, _ = run_window(mk, tk, 10, 100, m, ab, profiled=False)      # the value: no per-pass events (they cost
                _, tab = run_window(mk, tk, 10, 100, m, ab)                     # ~14 % of a 1M-particle step), then the passes
                wl[name] = {"value": round(m / (t * 1e-3) / 1e6, 2), "ms_per_step": round(t, 4), "particles": m,
                            "passes_ms": {k: v["ms"] for k, v in tab.items()},
                            "step_frac_of_hbm_peak": round(at * m / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                d3 = max(tab, key=lambda k: tab[k]["ms"])       # this workload's own dominant pass, bound from ITS counters
                b3, c3 = bound_from_evidence(d3, tab[d3]["alg_GBps"] / HBM_COPY_GBS, counters, name.startswith("dam_break_3d"),
                                             PASS_KERNEL_3D_TOL if name.endswith("tolerance_math") else None)
                wl[name]["roofline"] = {"bound": b3, "kernel": d3, "achieved": tab[d3]["alg_GBps"], "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": tab[d3]["frac"], "alg_bytes_per_particle": ab[d3],
                                        "valu_issue_frac": c3.get("valu_issue_frac") if c3 else None,
                                        "counter_kernel": c3.get("kernel") if c3 else None, "traffic": None}
            except Exception as e:      # an extra must never take the headline down with it
                wl[name] = {"error": str(e)}
        out["alt_workloads"] = dict(wl, note="10 warm-up + 100 timed steps each, strict mode unless named otherwise (+tolerance_math = fs3_create_ex "
                                            "FS_MATH_TOLERANCE: rtol 1e-5 per step vs the 3D oracle, keys bit-exact), one GPU; 3D has no reference counterpart")
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
