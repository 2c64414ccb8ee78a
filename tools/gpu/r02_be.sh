#!/bin/bash
set -o pipefail
O=gpurun_out/r02be; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python3 bench.py --no-build --no-alt --no-cpu-baseline --workload dam_break_2d_1M > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows[:24]:
    per=float(r['TotalDurationNs'])/220e3
    tot+=per
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), 'us', ('%.1f'%per).rjust(8), 'us/step')
print('sum', tot)
PY
tail -1 $O/stats.log | cut -c1-300
