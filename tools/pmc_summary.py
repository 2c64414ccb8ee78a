"""Mean counter values per dispatch of kernels matching a substring, from rocprofv3 counter_collection CSVs.
usage: python tools/pmc_summary.py <dir> <kernel-substring> [last_n]"""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 10
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat in row["Kernel_Name"]:
                per[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(per.items()):
        v = v[-last:]
        print(f"{k:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
