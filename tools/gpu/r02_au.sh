#!/bin/bash
set -o pipefail
O=gpurun_out/r02au; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_sort_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for lib in libfs_l3.so default libfs_l3.so default; do
  timeout -k 10 200 python tools/ab_mode.py strict bitonic 10 100 $lib >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done
cut -c1-150 $O/ab.txt
