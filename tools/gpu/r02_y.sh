#!/bin/bash
set -o pipefail
O=gpurun_out/r02y; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_3d.py tests/test_multi_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for w in "5 20" "10 100" "150 40" "5 20"; do set -- $w; python tools/ab_mode.py strict bitonic $1 $2 >> $O/ab.txt 2>&1; done
cat $O/ab.txt
