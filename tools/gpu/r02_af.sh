#!/bin/bash
set -o pipefail
O=gpurun_out/r02af; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_sort_gpu.py tests/test_3d.py tests/test_multi_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { echo "$*" >> $O/ab.txt; env "$@" timeout -k 10 300 python $T >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }; }
T="tools/ab_3d.py 10 100"
run FS_SORT_FUSE_STAGE=0
run FS_X=1
run FS_SORT_FUSE_STAGE=18
run FS_SORT_FUSE_STAGE=19
T="tools/ab_mode.py strict bitonic 10 100"
run FS_X=1
cat $O/ab.txt
