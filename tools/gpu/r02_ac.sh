#!/bin/bash
set -o pipefail
O=gpurun_out/r02ac; mkdir -p $O; rm -f $O/ab.txt
for c in 0 1024 256 0 1024; do
  echo "FUSE=16 EXP_COMPACT=$c (timing only)" >> $O/ab.txt
  FS_SORT_FUSE_STAGE=16 FS_SORT_EXP_COMPACT=$c python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done
cat $O/ab.txt
