#!/bin/bash
set -o pipefail
O=gpurun_out/r02i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "tolerance" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; }
tail -5 $O/pytest.log
for v in default build/libfs_noaq.so build/libfs_nokey.so build/libfs_neither.so default; do
  python tools/ab_variant.py $v 10 100 >> $O/ab.txt 2>&1 || exit 1
done
python tools/ab_mode.py tol bitonic 10 100 >> $O/ab.txt 2>&1
python tools/ab_mode.py tol counting 10 100 >> $O/ab.txt 2>&1
python tools/ab_mode.py ulp bitonic 10 100 >> $O/ab.txt 2>&1
python tools/ab_mode.py tol bitonic 150 40 >> $O/ab.txt 2>&1
cat $O/ab.txt
