#!/bin/bash
set -o pipefail
O=gpurun_out/r02ba; mkdir -p $O; rm -f $O/slab.txt
timeout -k 10 1100 python -m pytest tests/test_multi_gpu.py tests/test_parity_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for w in 8 2; do timeout -k 10 200 python tools/slab_overhead.py 16777216 $w >> $O/slab.txt 2>&1 || { tail -5 $O/slab.txt; exit 1; }; done
timeout -k 10 200 python tools/ab_mode.py strict counting 10 100 >> $O/slab.txt 2>&1
cut -c1-220 $O/slab.txt
