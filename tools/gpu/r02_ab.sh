#!/bin/bash
set -o pipefail
O=gpurun_out/r02ab; mkdir -p $O; rm -f $O/ab.txt
for f in 0 17 18 19 20; do
  echo "FS_SORT_FUSE_STAGE=$f" >> $O/ab.txt
  FS_SORT_FUSE_STAGE=$f python tools/ab_mode.py strict bitonic 150 100 >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done
for f in 18 19; do
  echo "FS_SORT_FUSE_STAGE=$f" >> $O/ab.txt
  FS_SORT_FUSE_STAGE=$f python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1
done
cat $O/ab.txt
