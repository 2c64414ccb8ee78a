#!/bin/bash
# single-GPU rehearsal of the N-rank bench path: gloo transport, every rank on device 0 (the nccl path needs N GPUs).
# Uses the bare launcher form (`python3 bench.py --gpus N`): bench.py starts its own ranks.
set -o pipefail
O=gpurun_out/${1:-rehearse}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build_product()" || exit 1
for n in 2 4; do
FS_DIST_BACKEND=gloo FS_FORCE_DEVICE0=1 FS_NO_SCALING_BASE=1 timeout -k 10 500 python3 bench.py --gpus $n --steps 10 --warmup 3 --no-build > $O/bench_$n.json 2> $O/bench_$n.err || { tail -20 $O/bench_$n.err; exit 1; }
tail -1 $O/bench_$n.json | cut -c1-900
done
