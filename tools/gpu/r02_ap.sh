#!/bin/bash
set -o pipefail
O=gpurun_out/r02ap; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_sort_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { echo "$*" >> $O/ab.txt; env "$@" timeout -k 10 200 python tools/ab_mode.py strict bitonic $W >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }; }
W="10 100"
run FS_SORT_MMAX_SHIFTED=4
run FS_SORT_MMAX_SHIFTED=5
run FS_SORT_MMAX_SHIFTED=6
run FS_SORT_MMAX_SHIFTED=4
run FS_SORT_MMAX_SHIFTED=5
W="150 100"
run FS_SORT_MMAX_SHIFTED=4
run FS_SORT_MMAX_SHIFTED=6
cat $O/ab.txt
