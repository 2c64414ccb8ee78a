#!/bin/bash
set -o pipefail
O=gpurun_out/r02s; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_multi_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1
python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1
python tools/ab_mode.py strict bitonic 150 40 >> $O/ab.txt 2>&1
cat $O/ab.txt
