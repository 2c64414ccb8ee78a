"""HBM traffic per kernel and per pass from two rocprofv3 PMC runs of the same command.

  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o p --output-format csv -- python3 bench.py --no-build --steps 10 --warmup 10 --no-alt --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o p --output-format csv -- python3 bench.py ... (same)
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write <2*(steps+warmup): bench.py runs the window twice, once with per-pass events> profiles/rNN_X_pmc_traffic_raw.json profiles/traffic_latest.json [commit]
(build first with `python __graft_entry__.py`: bench.py --no-build never compiles inside a profiled process)

HBM bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: on gfx950 FETCH_SIZE reports half of wide reads
(MI355X_MICROARCH.md, HBM section); calibrated here on k_import_aos, which reads 32 B x N.
"""
def _head():
    import subprocess
    try:
        return subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, timeout=10).stdout.strip() or 'unknown (no git on this box: pass it as argv[6])'
    except Exception:
        return 'unknown (no git on this box: pass it as argv[6])'


import collections, csv, glob, json, re, sys

def per_kernel(d, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
                tot[name] += float(row["Counter_Value"])
                cnt[name] += 1
    return tot, cnt

def pass_of(name):
    if "k_reorder" in name or "k_fill_gaps" in name or "k_cs_fixreorder" in name: return "reorder"
    if "bitonic" in name or "k_late_" in name or "k_cs_" in name or "csort" in name or "k_scan_lookback" in name: return "sort"
    if "k_density" in name: return "density"
    if "k_force" in name: return "force"
    return None

fetch_dir, write_dir, steps, raw_out, latest_out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
ft, fc = per_kernel(fetch_dir, "FETCH_SIZE")
wt, wc = per_kernel(write_dir, "WRITE_SIZE")
raw, per_launch = {}, {}
by_pass = collections.defaultdict(float)
for k in ft:
    launches = fc[k]
    raw[k] = {"fetch_KB_per_launch": ft[k] / launches, "launches": launches,
              "write_KB_per_launch": wt.get(k, 0.0) / max(wc.get(k, 1), 1)}
    b = 2 * raw[k]["fetch_KB_per_launch"] * 1024 + raw[k]["write_KB_per_launch"] * 1024
    per_launch[k] = int(b)
    p = pass_of(k)
    if p:
        by_pass[p] += b * launches / steps
json.dump(raw, open(raw_out, "w"), indent=1)
latest = {
    "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), bench.py --steps 10 --warmup 10 --no-alt, "
              "dam_break_2d_16M; HBM bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE reports half of wide "
              "reads: MI355X_MICROARCH.md HBM section; check: k_import_aos reads 32 B x N). predict+key is fused into "
              "the sort's first kernel. Produced by tools/pmc_traffic.py from " + raw_out + ".",
    "window": "rocprofv3 --pmc passes over bench.py --steps 10 --warmup 10 (both of its runs of the window, all 40 steps averaged), commit " + (sys.argv[6] if len(sys.argv) > 6 else _head()),
    "bytes_per_step_by_pass": {"predict_key": 0, **{p: int(by_pass[p]) for p in ("sort", "reorder", "density", "force")}},
    "bytes_per_launch": per_launch,
    "particles": 1 << 24,
}
json.dump(latest, open(latest_out, "w"), indent=1)
tot = sum(latest["bytes_per_step_by_pass"].values())
print({p: round(v / (1 << 24), 1) for p, v in latest["bytes_per_step_by_pass"].items()}, "B/particle; step total", round(tot / (1 << 24), 1))
