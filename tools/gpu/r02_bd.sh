#!/bin/bash
set -o pipefail
O=gpurun_out/r02bd; mkdir -p $O; rm -f $O/ab.txt
for lib in default libfs_eu4.so libfs_eu3.so default libfs_eu4.so; do
  timeout -k 10 200 python tools/ab_mode.py strict bitonic 10 100 $lib >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done
cut -c1-130 $O/ab.txt
