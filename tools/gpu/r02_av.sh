#!/bin/bash
set -o pipefail
O=gpurun_out/r02av; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_sort_gpu.py tests/test_parity_gpu.py tests/test_3d.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for lib in libfs_gb4.so default libfs_gb4.so default; do
  timeout -k 10 200 python tools/ab_mode.py strict bitonic 10 100 $lib >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done
for lib in libfs_gb4.so default; do
  timeout -k 10 200 python tools/ab_mode.py strict bitonic 5 20 $lib >> $O/ab.txt 2>&1
  timeout -k 10 200 python tools/ab_mode.py strict bitonic 150 100 $lib >> $O/ab.txt 2>&1
done
cut -c1-150 $O/ab.txt
