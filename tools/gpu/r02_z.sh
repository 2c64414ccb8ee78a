#!/bin/bash
# A/B: the block's own {vel, rho2} records read from LDS in the lean force kernel (FS_OWN_LDS)
set -o pipefail
O=gpurun_out/r02z; mkdir -p $O; rm -f $O/ab.txt
for lib in default libfs_own.so default libfs_own.so; do
  python tools/ab_mode.py strict bitonic 10 100 $lib >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done
for lib in default libfs_own.so; do
  python tools/ab_mode.py tol bitonic 10 100 $lib >> $O/ab.txt 2>&1
  python tools/ab_mode.py strict bitonic 150 40 $lib >> $O/ab.txt 2>&1
done
cat $O/ab.txt
