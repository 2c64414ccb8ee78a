#!/bin/bash
set -o pipefail
O=gpurun_out/r02bh; mkdir -p $O; rm -f $O/ab.txt
for n in 4194304 2097152 8388608; do for gb in 4 3; do
FS_SORT_GB=$gb timeout -k 10 200 python tools/ab_n.py $n >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done; done
cat $O/ab.txt
