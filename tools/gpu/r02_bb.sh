#!/bin/bash
set -o pipefail
O=gpurun_out/r02bb; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_3d.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 600 python tools/long_validation3d.py > $O/val3d.txt 2>&1 || { tail -8 $O/val3d.txt; exit 1; }
tail -3 $O/val3d.txt
timeout -k 10 300 python tools/ab_3d.py 10 100 >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
timeout -k 10 300 python tools/ab_3d.py 10 40 >> $O/ab.txt 2>&1
cat $O/ab.txt
