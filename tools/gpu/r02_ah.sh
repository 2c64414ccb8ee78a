#!/bin/bash
set -o pipefail
O=gpurun_out/r02ah; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python3 tools/ab_3d.py 10 100 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
head -12 $O/kernel_stats.csv | cut -c1-200
tail -2 $O/stats.log
