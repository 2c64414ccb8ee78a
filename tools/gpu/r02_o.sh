#!/bin/bash
set -o pipefail
O=gpurun_out/r02o; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1
python tools/ab_mode.py tol bitonic 10 100 >> $O/ab.txt 2>&1
python tools/ab_mode.py strict bitonic 150 40 >> $O/ab.txt 2>&1
python tools/ab_mode.py tol bitonic 150 40 >> $O/ab.txt 2>&1
python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1
cat $O/ab.txt
