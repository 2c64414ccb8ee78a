#!/bin/bash
O=gpurun_out/r02x; mkdir -p $O; rm -f $O/ab.txt
for lib in default build/libfs_g5.so build/libfs_g5s.so; do
  for w in "5 20" "10 100" "150 40"; do set -- $w; python tools/ab_mode.py strict bitonic $1 $2 $lib >> $O/ab.txt 2>&1; done
done
echo "--- late-form strided passes (default lib)" >> $O/ab.txt
for f in 99 20 18 16 15; do
  echo "FS_SORT_LATE_FORM=$f" >> $O/ab.txt
  FS_SORT_LATE_FORM=$f python tools/ab_mode.py strict bitonic 10 100 default >> $O/ab.txt 2>&1
done
FS_SORT_LATE_FORM=16 timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q >> $O/ab.txt 2>&1
cat $O/ab.txt
