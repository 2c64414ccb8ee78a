#!/bin/bash
set -o pipefail
O=gpurun_out/r02ar; mkdir -p $O; rm -f $O/ab.txt
for w in "5 20" "10 100" "150 100" "5 20"; do set -- $w; timeout -k 10 200 python tools/ab_mode.py strict bitonic $1 $2 >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }; done
cat $O/ab.txt
