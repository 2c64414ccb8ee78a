#!/bin/bash
set -o pipefail
O=gpurun_out/r02h; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_multi_gpu.py tests/test_export_gpu.py tests/test_field.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/ab_variant.py default 10 100 >> $O/ab.txt 2>&1 || exit 1
python tools/ab_variant.py default 150 40 >> $O/ab.txt 2>&1 || exit 1
cat $O/ab.txt
