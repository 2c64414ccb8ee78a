#!/bin/bash
O=gpurun_out/r02u; mkdir -p $O; rm -f $O/ab.txt
for lib in default build/libfs_pipe2.so default build/libfs_pipe2.so; do
  python tools/ab_mode.py strict bitonic 10 100 $lib >> $O/ab.txt 2>&1
  python tools/ab_mode.py tol bitonic 10 100 $lib >> $O/ab.txt 2>&1
done
python tools/ab_mode.py strict bitonic 150 40 default >> $O/ab.txt 2>&1
python tools/ab_mode.py strict bitonic 150 40 build/libfs_pipe2.so >> $O/ab.txt 2>&1
cat $O/ab.txt
