#!/bin/bash
set -o pipefail
O=gpurun_out/r02al; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_sort_gpu.py tests/test_parity_gpu.py tests/test_3d.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { echo "$*" >> $O/ab.txt; env "$@" timeout -k 10 200 python tools/ab_mode.py strict bitonic $W >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }; }
W="10 100"
run FS_SORT_FUSED12=0
run FS_SORT_FUSED12=1
run FS_SORT_FUSED12=0
run FS_SORT_FUSED12=1
W="5 20"
run FS_SORT_FUSED12=0
run FS_SORT_FUSED12=1
W="150 100"
run FS_SORT_FUSED12=0
run FS_SORT_FUSED12=1
cat $O/ab.txt
