"""Timeline of ONE step out of a `rocprofv3 --kernel-trace ... --output-format csv` run: every dispatch between two
consecutive launches of a delimiter kernel ON ONE QUEUE, with start offset, duration and the queue (= HIP stream) it ran on —
shows what runs beside what (the halo exchange beside the interior columns' kernels in an overlapped slab step).

  python tools/trace_timeline.py DIR/p_kernel_trace.csv [delimiter=k_slab_pack] [which=-2] [queue-of-delimiter index=middle]
"""
import collections
import csv
import re
import sys

path = sys.argv[1]
delim = sys.argv[2] if len(sys.argv) > 2 else "k_slab_pack"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("fsd::", "").strip()


rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
rows.sort()
byq = collections.defaultdict(list)
allm = []
for k, (s, e, n, q) in enumerate(rows):
    if delim in n:
        byq[q].append(k)
        allm.append((k, q))
if len(sys.argv) > 4:
    # the queue with the most delimiter launches, ties -> the one whose launches come second in time (the middle rank of tools/slab_overhead.py)
    qs = sorted(byq, key=lambda q: (-len(byq[q]), rows[byq[q][0]][0]))
    q = qs[int(sys.argv[4])]
    marks = byq[q]
    a = marks[which]
else:
    # `which` counts delimiter launches over ALL queues in time order (negative: from the end of the run, where only the measured
    # handle is still stepping); the step shown runs to the next delimiter launch on the same queue
    a, q = allm[which]
    marks = byq[q]
nxt = [m for m in marks if m > a]
b = nxt[0] if nxt else len(rows)
t0 = rows[a][0]
tend = rows[b][0] if b < len(rows) else rows[-1][1]
print(f"step of queue {q}: {delim} launch #{which} .. the next one; offsets in us from its start")
for s, e, n, qq in rows[a:]:
    if s >= tend:
        break
    mark = "*" if qq == q else " "
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{qq:>3}{mark} {n[:90]}")
