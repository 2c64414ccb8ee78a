#!/bin/bash
set -o pipefail
O=gpurun_out/r02g; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -12 $O/pytest_gpu.log
