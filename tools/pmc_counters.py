"""Per-kernel SQ/GRBM counter summary from one or more `rocprofv3 --pmc` passes of tools/pmc_run.py.

  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES ... --kernel-trace -d gpurun_out/pmcA -o p --output-format csv -- python3 tools/pmc_run.py 2d 10 20
  python tools/pmc_counters.py <tag> <warm> <steps> <out_prefix> <dirA> [<dirB> ...] [--latest profiles/counters_latest.json] [--note "..."]

Every pass dispatches the same kernels in the same order; a kernel's dispatches of the timed part (the last
steps/(warm+steps) of them) are averaged per counter.  Derived columns (chip: 256 CUs x 4 SIMDs = 1024 SIMDs):
  cycles          = GRBM_GUI_ACTIVE / 8            (rocprofv3 sums the counter over the 8 XCDs; MI355X guide, DVFS note)
  valu_issue_frac = ((SQ_INSTS_VALU - TRANS) * 2.30 + TRANS * 4.56) / (cycles * 1024)   — share of the VALU issue slots in use.
                    2.30 / 4.56 = measured cycles per wave-instruction per SIMD at 8 waves/SIMD for plain ops / v_rcp,v_sqrt,v_rsq
                    (tools/valu_issue_table.hip, profiles/r02_a_valu_issue_table.txt).  SQ_ACTIVE_INST_VALU is NOT a busy time on
                    gfx950: it reads 1 quad-cycle (4 cycles) per instruction whatever the instruction (valu_cyc_per_inst column),
                    so SQ_ACTIVE_INST_VALU * 4 / cycles over-reads the issue share by 4 / 2.3 (it exceeds 1 for dense kernels).
  waves_per_simd  = SQ_WAVE_CYCLES * 4 / (cycles * 1024)          (achieved occupancy, of 8)
  valu_insts_per_wave = SQ_INSTS_VALU / SQ_WAVES
  valu_cyc_per_inst   = SQ_ACTIVE_INST_VALU * 4 / SQ_INSTS_VALU   (issue cycles a wave-instruction holds the SIMD)
  lds_busy_frac   = SQ_LDS_IDX_ACTIVE / (cycles * 256)  (LDS array cycles per CU), bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_frac       = SQ_WAIT_ANY / SQ_WAVE_CYCLES  (waves parked on s_waitcnt / barriers), wait_inst_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  us              = mean (End - Start) of the dispatch under the profiler (counter runs serialise kernels: read shares, not totals)
"""
import collections, csv, glob, json, re, sys

args = sys.argv[1:]
latest = note = None
if "--latest" in args:
    i = args.index("--latest"); latest = args[i + 1]; del args[i:i + 2]
if "--note" in args:
    i = args.index("--note"); note = args[i + 1]; del args[i:i + 2]
tag, warm, steps, out_prefix, dirs = args[0], int(args[1]), int(args[2]), args[3], args[4:]

def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()

vals = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> [values in dispatch order]
meta = {}
for d in dirs:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        seen = collections.defaultdict(dict)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                did = int(row["Dispatch_Id"])
                seen[(k, did)][row["Counter_Name"]] = float(row["Counter_Value"])
                seen[(k, did)]["us"] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                meta[k] = {"vgpr": int(row["VGPR_Count"]), "sgpr": int(row["SGPR_Count"]), "lds": int(row["LDS_Block_Size"]),
                           "wg": int(row["Workgroup_Size"])}
        for (k, did) in sorted(seen, key=lambda t: t[1]):
            for c, v in seen[(k, did)].items():
                vals[k][c].append(v)

frac = steps / float(warm + steps)
rows = {}
for k, per in vals.items():
    if not k.startswith("fsd::") and not k.startswith("k3") and "k_" not in k:
        continue
    r = {}
    for c, v in per.items():
        keep = max(1, int(round(len(v) * frac)))
        v = v[-keep:]
        r[c] = sum(v) / len(v)
    r["dispatches_averaged"] = max(1, int(round(len(next(iter(per.values()))) * frac)))
    cyc = r.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    g = lambda name: r.get(name)
    d = dict(meta.get(k, {}))
    d["us"] = round(r.get("us", 0.0), 1)
    d["cycles"] = round(cyc)
    if cyc > 0:
        if g("SQ_INSTS_VALU") is not None:
            tr = g("SQ_INSTS_VALU_TRANS_F32") or 0.0
            d["valu_issue_frac"] = round(((g("SQ_INSTS_VALU") - tr) * 2.30 + tr * 4.56) / (cyc * 1024), 3)
            d["valu_ginst_per_s_per_simd"] = round(g("SQ_INSTS_VALU") / 1024 / max(r.get("us", 0.0), 1e-9) / 1e3, 3)
        if g("SQ_WAVE_CYCLES") is not None: d["waves_per_simd"] = round(g("SQ_WAVE_CYCLES") * 4 / (cyc * 1024), 2)
        if g("SQ_LDS_IDX_ACTIVE") is not None: d["lds_busy_frac"] = round(g("SQ_LDS_IDX_ACTIVE") / (cyc * 256), 3)
        if g("SQ_ACTIVE_INST_LDS") is not None: d["lds_inst_busy_frac"] = round(g("SQ_ACTIVE_INST_LDS") * 4 / (cyc * 1024), 3)
        if g("SQ_ACTIVE_INST_VMEM") is not None: d["vmem_inst_busy_frac"] = round(g("SQ_ACTIVE_INST_VMEM") * 4 / (cyc * 1024), 3)
        if g("SQ_ACTIVE_INST_SCA") is not None: d["salu_busy_frac"] = round(g("SQ_ACTIVE_INST_SCA") * 4 / (cyc * 1024), 3)
    if g("SQ_WAVES"):
        for c, nm in (("SQ_INSTS_VALU", "valu_insts_per_wave"), ("SQ_INSTS_SALU", "salu_insts_per_wave"),
                      ("SQ_INSTS_LDS", "lds_insts_per_wave"), ("SQ_INSTS_VMEM_RD", "vmem_rd_insts_per_wave"),
                      ("SQ_INSTS_VALU_TRANS_F32", "trans_insts_per_wave"), ("SQ_INSTS_VALU_FMA_F32", "fma_insts_per_wave"),
                      ("SQ_INSTS_VALU_MUL_F32", "mul_insts_per_wave"), ("SQ_INSTS_VALU_ADD_F32", "add_insts_per_wave"),
                      ("SQ_INSTS_VALU_INT32", "int32_insts_per_wave"), ("SQ_INSTS_VALU_CVT", "cvt_insts_per_wave")):
            if g(c) is not None: d[nm] = round(g(c) / g("SQ_WAVES"), 1)
        d["waves"] = round(g("SQ_WAVES"))
    if g("SQ_INSTS_VALU") and g("SQ_ACTIVE_INST_VALU") is not None:
        d["valu_cyc_per_inst"] = round(g("SQ_ACTIVE_INST_VALU") * 4 / g("SQ_INSTS_VALU"), 2)
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        d["valu_lane_util"] = round(g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64), 3)
    if g("SQ_WAVE_CYCLES"):
        if g("SQ_WAIT_ANY") is not None: d["wait_frac"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3)
        if g("SQ_WAIT_INST_ANY") is not None: d["wait_inst_frac"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 3)
        if g("SQ_ACTIVE_INST_ANY") is not None: d["active_inst_frac"] = round(g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 3)
    if g("SQ_LDS_IDX_ACTIVE"):
        if g("SQ_LDS_BANK_CONFLICT") is not None: d["bank_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 3)
    d["raw"] = {c: round(v, 1) for c, v in r.items() if c.isupper() or c.startswith("SQ") or c.startswith("GRBM")}
    rows[k] = d

src = f"rocprofv3 --pmc (SQ/GRBM passes: {', '.join(dirs)}) -- python3 tools/pmc_run.py {tag}; warm {warm} + {steps} steps, last {steps} steps' dispatches averaged; tools/pmc_counters.py" + (f"; {note}" if note else "")
cols = ["us", "cycles", "waves", "vgpr", "lds", "waves_per_simd", "valu_issue_frac", "valu_ginst_per_s_per_simd", "valu_insts_per_wave", "valu_cyc_per_inst", "valu_lane_util",
        "salu_insts_per_wave", "lds_insts_per_wave", "vmem_rd_insts_per_wave", "trans_insts_per_wave", "lds_busy_frac", "bank_conflict_frac",
        "wait_frac", "wait_inst_frac", "active_inst_frac"]
order = sorted(rows, key=lambda k: -rows[k].get("us", 0) * rows[k]["raw"].get("dispatches_averaged", 1))
with open(out_prefix + ".csv", "w") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel"] + cols + ["raw_counters_json"])
    for k in order:
        w.writerow([k] + [rows[k].get(c, "") for c in cols] + [json.dumps(rows[k]["raw"])])
with open(out_prefix + ".md", "w") as fh:
    fh.write(f"# counters — {tag}\n\n{src}\n\n" + __doc__.split("Derived columns")[1].join(["Derived columns", ""]) + "\n")
    fh.write("| kernel | " + " | ".join(cols) + " |\n|---|" + "---|" * len(cols) + "\n")
    for k in order:
        fh.write(f"| `{k}` | " + " | ".join(str(rows[k].get(c, "")) for c in cols) + " |\n")
if latest:
    try:
        cur = json.load(open(latest))
    except (OSError, ValueError):
        cur = {"kernels": {}}
    # every kernel row names the run it came from (bench.py quotes roofline.valu_issue.source per kernel: a 2D kernel must never be
    # attributed to a 3D run); the file-level "source" only lists the runs that contributed
    runs = [r for r in cur.get("source", "").split(" || ") if r]
    if src not in runs:
        runs.append(src)
    cur["source"] = " || ".join(runs)
    for k, d in rows.items():
        cur["kernels"][k] = dict({c: d[c] for c in cols if c in d}, source=src)
    json.dump(cur, open(latest, "w"), indent=1)
for k in order[:12]:
    print(k, {c: rows[k].get(c) for c in ("us", "waves_per_simd", "valu_issue_frac", "valu_ginst_per_s_per_simd", "valu_insts_per_wave", "valu_cyc_per_inst", "lds_busy_frac", "wait_frac")})
