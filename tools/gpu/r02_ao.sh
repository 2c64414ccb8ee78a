#!/bin/bash
set -o pipefail
O=gpurun_out/r02ao; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "tolerance" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for lib in default libfs_h8.so libfs_h4.so; do
  timeout -k 10 200 python tools/ab_mode.py tol bitonic 10 100 $lib >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done
FS_TOL_TILE=0 timeout -k 10 200 python tools/ab_mode.py tol bitonic 10 100 >> $O/ab.txt 2>&1
cat $O/ab.txt
