#!/bin/bash
set -o pipefail
O=gpurun_out/r02ay; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for i in 1 2; do timeout -k 10 200 python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }; done
timeout -k 10 200 python tools/ab_mode.py strict counting 10 100 >> $O/ab.txt 2>&1
cut -c1-150 $O/ab.txt
