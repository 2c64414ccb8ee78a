#!/bin/bash
set -o pipefail
O=gpurun_out/r02aw; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for lib in default libfs_gb4.so; do
rocprofv3 --kernel-trace --stats -d $O/stats_$lib -o p --output-format csv -- python3 tools/ab_mode.py strict bitonic 10 100 $lib > $O/stats_$lib.log 2>&1 || { tail -5 $O/stats_$lib.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$O/stats_$lib/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("== $lib")
for r in rows[:14]:
    if 'bitonic' in r['Name'] or 'late' in r['Name']:
        print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), 'us', ('%.1f'%(float(r['TotalDurationNs'])/110e3)).rjust(8), 'us/step')
PY
done
