#!/bin/bash
set -o pipefail
O=gpurun_out/r02c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_constdiv_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1 || { tail -40 $O/pytest_parity.log; exit 1; }
tail -3 $O/pytest_parity.log
for v in default build/libfs_w7.so build/libfs_w6.so build/libfs_w5.so default; do
  python tools/ab_variant.py $v 10 100 >> $O/ab.txt 2>&1 || exit 1
done
for v in default build/libfs_w6.so build/libfs_w5.so; do
  python tools/ab_variant.py $v 150 40 >> $O/ab.txt 2>&1 || exit 1
done
cat $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_multi_gpu.py -m gpu -x -q --durations=5 > $O/pytest_multi.log 2>&1 || { tail -40 $O/pytest_multi.log; exit 1; }
tail -12 $O/pytest_multi.log
