#!/bin/bash
set -o pipefail
O=gpurun_out/r02t; mkdir -p $O
python -c "import __graft_entry__ as g; g.build_checker()" || exit 1
timeout -k 10 1000 python tools/long_validation.py 262144 500 100 > $O/long_validation.txt 2>&1 || { tail -20 $O/long_validation.txt; exit 1; }
tail -12 $O/long_validation.txt
timeout -k 10 600 python tools/fuzz_parity.py 1000 150 > $O/fuzz_parity.txt 2>&1 || { tail -20 $O/fuzz_parity.txt; exit 1; }
tail -3 $O/fuzz_parity.txt
timeout -k 10 300 python tools/fuzz_api.py 2000 100 > $O/fuzz_api.txt 2>&1 || { tail -20 $O/fuzz_api.txt; exit 1; }
tail -2 $O/fuzz_api.txt
timeout -k 10 400 python tools/fuzz_slabs.py 500 60 > $O/fuzz_slabs.txt 2>&1 || { tail -20 $O/fuzz_slabs.txt; exit 1; }
tail -2 $O/fuzz_slabs.txt
