#!/bin/bash
set -o pipefail
O=gpurun_out/r02q; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_multi_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for c in 13 10 8 6 4; do
  echo "chunk_log2 $c" >> $O/ab.txt
  FS_XCD_CHUNK_LOG2=$c python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1
  FS_XCD_CHUNK_LOG2=$c python tools/ab_mode.py strict bitonic 150 40 >> $O/ab.txt 2>&1
done
cat $O/ab.txt
