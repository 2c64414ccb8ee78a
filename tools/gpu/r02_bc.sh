#!/bin/bash
set -o pipefail
O=gpurun_out/r02bc; mkdir -p $O; rm -f $O/ab.txt
for lib in libfs_w4.so default libfs_w6.so libfs_w4.so; do
  timeout -k 10 300 python tools/ab_3d.py 10 100 $lib >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done
cat $O/ab.txt
