#!/bin/bash
set -o pipefail
O=gpurun_out/r02as; mkdir -p $O; rm -f $O/ab.txt
for lib in default libfs_g2k.so libfs_g4k.so default libfs_g2k.so; do
  timeout -k 10 200 python tools/ab_mode.py strict bitonic 10 100 $lib >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done
for lib in default libfs_g2k.so libfs_g4k.so; do
  timeout -k 10 200 python tools/ab_mode.py strict bitonic 150 100 $lib >> $O/ab.txt 2>&1
done
cut -c1-160 $O/ab.txt
