#!/bin/bash
set -o pipefail
O=gpurun_out/r02bf; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_multi_gpu.py tests/test_export_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for w in "5 20" "10 100" "150 100"; do set -- $w; timeout -k 10 200 python tools/ab_mode.py strict bitonic $1 $2 >> $O/ab.txt 2>&1 || { tail -5 $O/ab.txt; exit 1; }; done
timeout -k 10 200 python tools/slab_overhead.py 16777216 8 >> $O/ab.txt 2>&1
python bench.py --no-build --no-alt --no-cpu-baseline --workload dam_break_2d_1M 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('1M', b['value'], b['ms_per_step'], {k:v['ms'] for k,v in b['roofline']['passes'].items()})" >> $O/ab.txt
cut -c1-200 $O/ab.txt
