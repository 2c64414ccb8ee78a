#!/bin/bash
# the two bench lines of the final code (the rest of the r02_k evidence: tools/gpu/r02_k.sh)
set -o pipefail
O=gpurun_out/r02k; mkdir -p $O
python -c "import __graft_entry__ as g; g.build_product(); g.build_checker()" || exit 1
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python bench.py --steps 20 --warmup 5 --no-alt > $O/bench_driver_window.json 2> $O/bench2.err || { tail -20 $O/bench2.err; exit 1; }
python - <<PY
import json
b=json.load(open("$O/bench.json")); print(b['value'], b['ms_per_step'])
for n,v in b['alt_workloads'].items():
    if isinstance(v,dict): print(n, v.get('value'), v.get('ms_per_step'))
d=json.load(open("$O/bench_driver_window.json")); print('driver', d['value'], d['ms_per_step'])
PY
