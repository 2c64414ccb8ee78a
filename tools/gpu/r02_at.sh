#!/bin/bash
set -o pipefail
O=gpurun_out/r02at; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python3 tools/ab_mode.py strict bitonic 10 100 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:24]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), 'us', ('%.1f'%(float(r['TotalDurationNs'])/110e3)).rjust(8), 'us/step')
PY
