#!/bin/bash
set -o pipefail
O=gpurun_out/r02e; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_3d.py -m gpu -x -q > $O/pytest_parity.log 2>&1 || { tail -40 $O/pytest_parity.log; exit 1; }
tail -3 $O/pytest_parity.log
python tools/ab_variant.py default 10 100 >> $O/ab.txt 2>&1 || exit 1
FS_SORT_MMAX=3 python tools/ab_variant.py default 10 100 >> $O/ab.txt 2>&1 || exit 1
FS_SORT_MMAX=5 python tools/ab_variant.py default 10 100 >> $O/ab.txt 2>&1 || exit 1
python tools/ab_variant.py default 150 40 >> $O/ab.txt 2>&1 || exit 1
cat $O/ab.txt
python bench.py --workload dam_break_2d_1M --no-alt --no-cpu-baseline > $O/bench_1m.json 2>$O/bench_1m.err || { tail $O/bench_1m.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_1m.json')); print('1M', d['ms_per_step'], d['roofline']['passes'])"
