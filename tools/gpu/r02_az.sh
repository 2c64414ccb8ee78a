#!/bin/bash
set -o pipefail
O=gpurun_out/r02az; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for w in 8 4 2; do timeout -k 10 200 python tools/slab_overhead.py 16777216 $w >> $O/slab.txt 2>&1 || { tail -5 $O/slab.txt; exit 1; }; done
cat $O/slab.txt
rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python3 tools/slab_overhead.py 16777216 8 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), 'us', ('%.1f'%(float(r['TotalDurationNs'])/60e3)).rjust(8), 'us/step')
PY
