#!/usr/bin/env python3
"""bench.py — headline benchmark: M particle-steps/s on the 16M-particle 2D dam break
(BASELINE.json `metric`, configs[2]; SURVEY.md §8d scene), one process per GPU.

A "step" is one pass of the hot path (predict -> key -> bitonic sort -> cell starts ->
density -> force+integrate) over all particles.  State is resident in HBM before the
timed region; timing uses HIP events on the simulation's own stream (C ABI
fs_timed_steps / fs_profile_*), bracketed by barrier + device sync, MAX over ranks.

Prints ONE JSON line on rank 0 (contract fields + `roofline` + `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md §8d — algorithmic (compulsory SoA) bytes per particle-step, by pass.
ALG_BYTES = {"predict_key": 28, "sort": 12, "reorder": 48 + 4, "density": 16, "force": 48}
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
ALG_BYTES_3D = {"predict_key": 40, "sort": 12, "reorder": 72 + 4, "density": 20, "force": 68}   # 216 B
ALG_TOTAL_3D = 216


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # >= 10 warm-up + >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="dam_break_2d_16M", choices=sorted(WORKLOADS))
    ap.add_argument("--sort", default="bitonic", choices=["bitonic", "counting"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra counting-sort measurement")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc run")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    if world > 1:
        # torch first: its bundled HIP runtime has the same SONAME as /opt/rocm's, so the engine
        # library binds to the one already loaded (one runtime per process).
        import torch  # noqa: F401
    import __graft_entry__ as ge
    ge.build()
    import gpu_fluid_simulation_amd as g

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
    sim.close()

    # roofline: the SAME window once more on a fresh handle, now with a HIP event at every pass boundary
    # (they serialise the kernel boundaries and cost ~1 %, which is why the headline window runs without them)
    sim = make_sim()
    for _ in range(args.warmup):
        sim.tick(tick)
    sim.sync()
    sim.profile(True)
    sim.profile_read(reset=True)
    ms_profiled = sim.timed_steps(tick, args.steps)
    sim.sync()
    passes, psteps = sim.profile_read(reset=True)
    assert psteps == args.steps
    sim.profile(False)

    ms_per_step = ms / args.steps
    value = n / (ms_per_step * 1e-3) / 1e6            # M particle-steps/s

    per_pass = {}
    for name, tot in passes.items():
        t = tot / args.steps
        gbs = alg_bytes[name] * n / (t * 1e-3) / 1e9 if t > 0 else 0.0
        per_pass[name] = {"ms": round(t, 4), "alg_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    dom = max(per_pass, key=lambda k: per_pass[k]["ms"])
    traffic = args.pmc_traffic
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if traffic is None and not is3d and args.workload == "dam_break_2d_16M" and os.path.exists(tpath):
        # HBM bytes per step of the dominant pass from the committed rocprofv3 --pmc runs (separate
        # FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction): profiles/traffic_latest.json
        traffic = json.load(open(tpath))["bytes_per_step_by_pass"].get(dom)
    roofline = {
        "bound": "hbm", "kernel": dom,
        "achieved": per_pass[dom]["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "profiled_ms_per_step": round(ms_profiled / args.steps, 4),
        "frac": per_pass[dom]["frac"],
        "traffic": traffic,
        "alg_bytes_per_particle": alg_bytes[dom],
        "step": {"alg_bytes_per_particle": alg_total,
                 "achieved": round(alg_total * n / (ms_per_step * 1e-3) / 1e9, 1),
                 "frac": round(alg_total * n / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "frac_of_measured_copy": round(alg_total * n / (ms_per_step * 1e-3) / 1e9 / HBM_COPY_GBS, 4)},
        "passes": per_pass,
    }
    out = {
        "metric": "M particle-steps/s", "value": round(value, 2), "unit": "M particle-steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "particles": n, "scene": "SURVEY.md §8d " + ("dam_break_3d (no reference counterpart)" if is3d else "dam_break_2d"),
                   "sort": args.sort, "ref_quirks": not is3d, "parallelism": "1 GPU"},
        "host_wall_ms_per_step": round(t_wall / args.steps, 4),
        "roofline": roofline,
    }
    sim.close()
    if not is3d and args.sort == "bitonic" and not args.no_alt:
        # extras (NOT the headline): the same scene and protocol in the engine's opt-in modes
        def alt_run(**kw):
            a = g.FluidSimulation(st, device=local_rank, initial_offset=off, **kw)
            for _ in range(args.warmup):
                a.tick(tick)
            a.sync()
            t = a.timed_steps(tick, args.steps) / args.steps
            a.close()
            return {"value": round(n / (t * 1e-3) / 1e6, 2), "unit": "M particle-steps/s", "ms_per_step": round(t, 4)}
        out["alt_modes"] = {
            "counting_sort": dict(alt_run(sort_mode=g.FS_SORT_COUNTING),
                                  note="stable O(N) cell sort instead of the reference network (SURVEY 8f-1); floats equal "
                                       "the headline path to summation-order tolerance"),
            "wgsl_ulp_math": dict(alt_run(math_mode=g.FS_MATH_WGSL_ULP),
                                  note="native rcp/sqrt in the force pass (<= ~1.5 ulp, inside WGSL's 2.5-ULP division "
                                       "contract for the reference shaders); not bit-exact vs the IEEE oracle"),
            "counting_sort+wgsl_ulp_math": alt_run(sort_mode=g.FS_SORT_COUNTING, math_mode=g.FS_MATH_WGSL_ULP),
        }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
