#!/bin/bash
# shifted late-stage merge: exactness tests, then A/B of the plan knob at 16M / 1M
set -o pipefail
O=gpurun_out/r02aa; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_sort_gpu.py tests/test_parity_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for f in 0 16 15 17 0 16; do
  echo "FS_SORT_FUSE_STAGE=$f" >> $O/ab.txt
  FS_SORT_FUSE_STAGE=$f python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done
for f in 0 16; do
  echo "FS_SORT_FUSE_STAGE=$f dense" >> $O/ab.txt
  FS_SORT_FUSE_STAGE=$f python tools/ab_mode.py strict bitonic 150 100 >> $O/ab.txt 2>&1
  FS_SORT_FUSE_STAGE=$f python tools/ab_mode.py strict bitonic 5 20 >> $O/ab.txt 2>&1
done
cat $O/ab.txt
