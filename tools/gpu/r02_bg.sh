#!/bin/bash
set -o pipefail
O=gpurun_out/r02bg; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_sort_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for wl in dam_break_2d_1M; do
for gb in 4 3 4 3; do
FS_SORT_GB=$gb python bench.py --no-build --no-alt --no-cpu-baseline --workload $wl 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('$wl gb=$gb', b['value'], b['ms_per_step'], {k:v['ms'] for k,v in b['roofline']['passes'].items()})" >> $O/ab.txt
done; done
cat $O/ab.txt
