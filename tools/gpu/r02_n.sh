#!/bin/bash
set -o pipefail
O=gpurun_out/r02n; mkdir -p $O; rm -f $O/ab.txt
for v in default build/libfs_fast.so default build/libfs_fast.so; do
  python tools/ab_variant.py $v 10 100 >> $O/ab.txt 2>&1 || exit 1
done
cat $O/ab.txt
