#!/bin/bash
# usage: prof_stats.sh <outdir-tag> [scene warm steps mode]  -> gpurun_out/<tag>/stats.csv (rocprofv3 --kernel-trace --stats)
set -o pipefail
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python3 tools/pmc_run.py "$@" > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp $F $O/stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats.csv")))
for r in rows[:16]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), 'us', r['Percentage'])
PY
