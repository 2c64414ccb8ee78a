#!/bin/bash
set -o pipefail
O=gpurun_out/r02w; mkdir -p $O; rm -f $O/ab.txt
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_multi_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python tools/ab_mode.py strict bitonic 10 100 >> $O/ab.txt 2>&1
python tools/ab_mode.py strict bitonic 150 40 >> $O/ab.txt 2>&1
cat $O/ab.txt
for w in dam_break_2d_1M dam_break_2d_4096 dam_break_3d_8M; do
  python bench.py --workload $w --no-alt --no-cpu-baseline > $O/bench_$w.json 2>$O/bench_$w.err || { tail $O/bench_$w.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_$w.json')); print('$w', d['ms_per_step'], d['value'], {k: v['ms'] for k, v in d['roofline']['passes'].items()})"
done
