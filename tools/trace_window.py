"""Per-kernel average duration over the LAST `steps` simulation steps of a `rocprofv3 --kernel-trace` run
(a whole-run --stats summary mixes the near-lattice start with the dense regime):

  rocprofv3 --kernel-trace -d DIR -o p --output-format csv -- python3 tools/pmc_run.py 2d 150 100
  python tools/trace_window.py DIR/p_kernel_trace.csv 100 [delimiter-kernel-substring]

A step is delimited by the dispatches of one kernel that runs exactly once per step (default: the density kernel)."""
import collections
import csv
import re
import sys

path, steps = sys.argv[1], int(sys.argv[2])
delim = sys.argv[3] if len(sys.argv) > 3 else "k_density"


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
marks = [k for k, (_, _, n) in enumerate(rows) if delim in n]
if len(marks) < steps + 1:
    raise SystemExit(f"only {len(marks)} dispatches of {delim}")
first = marks[-steps - 1] + 1            # right after the density launch of the step before the window
# a step runs sort .. force; start the window at the first kernel after that step's force: find the next delimiter's step start
lo = rows[first][0]
acc = collections.defaultdict(lambda: [0, 0.0])
t_first, t_last = None, None
for s, e, n in rows[first:marks[-1] + 1]:
    acc[n][0] += 1
    acc[n][1] += (e - s) / 1e3
    t_first = s if t_first is None else t_first
    t_last = e
tot = sum(v[1] for v in acc.values())
print(f"window: last {steps} steps (delimited by {delim}); sum of kernel time {tot / steps:.1f} us/step, wall {(t_last - t_first) / 1e3 / steps:.1f} us/step")
for n, (c, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:72]:72s} {c / steps:7.2f} /step {us / c:9.1f} us avg {us / steps:9.1f} us/step")
