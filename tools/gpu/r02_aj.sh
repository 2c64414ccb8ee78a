#!/bin/bash
# cost of the stand-by kernel when it has to work: a fixed stage that often fails, with / without the single stand-by launch
set -o pipefail
O=gpurun_out/r02aj; mkdir -p $O; rm -f $O/ab.txt
run() { echo "$*" >> $O/ab.txt; env "$@" timeout -k 10 200 python tools/ab_mode.py strict bitonic $W >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }; }
W="10 100"
run FS_SORT_FUSE_STAGE=16
run FS_SORT_FUSE_STAGE=16 FS_SORT_TRUST=1
run FS_SORT_FUSE_STAGE=15
run FS_SORT_FUSE_STAGE=15 FS_SORT_TRUST=1
run FS_SORT_FUSE_STAGE=17 FS_SORT_TRUST=1
cat $O/ab.txt
