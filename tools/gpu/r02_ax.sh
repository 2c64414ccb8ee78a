#!/bin/bash
set -o pipefail
O=gpurun_out/r02ax; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_constdiv_gpu.py tests/test_parity_gpu.py tests/test_multi_gpu.py tests/test_sort_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { echo "$*" >> $O/ab.txt; env "$@" timeout -k 10 200 python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }; }
run FS_NO_CONSTDIV=1
run FS_X=1
run FS_NO_CONSTDIV=1
run FS_X=1
cut -c1-150 $O/ab.txt
