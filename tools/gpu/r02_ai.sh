#!/bin/bash
set -o pipefail
O=gpurun_out/r02ai; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_3d.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for lib in default libfs_w6.so; do
  timeout -k 10 300 python tools/ab_3d.py 10 100 $lib >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }
done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python3 tools/ab_3d.py 10 100 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
