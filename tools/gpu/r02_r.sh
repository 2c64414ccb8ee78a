#!/bin/bash
O=gpurun_out/r02r; mkdir -p $O; rm -f $O/ab.txt
for cfg in "4 18" "5 18" "6 18" "6 16" "5 16" "6 20"; do
  set -- $cfg
  echo "late mmax $1 from stage $2" >> $O/ab.txt
  FS_SORT_MMAX_LATE=$1 FS_SORT_LATE_STAGE=$2 python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1
done
FS_SORT_MMAX_LATE=6 FS_SORT_LATE_STAGE=18 timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "sort or 16m or 64m" >> $O/ab.txt 2>&1
cat $O/ab.txt
