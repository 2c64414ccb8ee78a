#!/bin/bash
O=gpurun_out/r02v; mkdir -p $O; rm -f $O/ab.txt
for lib in default build/libfs_dw.so default build/libfs_dw.so; do
  python tools/ab_mode.py strict bitonic 10 100 $lib >> $O/ab.txt 2>&1
done
cat $O/ab.txt
