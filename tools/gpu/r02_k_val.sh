#!/bin/bash
# round-2 validation campaigns on the final code: long runs against the oracle, fuzz of parity / API / slabs
set -o pipefail
O=gpurun_out/r02k; mkdir -p $O
python -c "import __graft_entry__ as g; g.build_product(); g.build_checker()" || exit 1
timeout -k 10 900 python tools/long_validation.py 262144 500 100 > $O/validation.txt 2>&1 || { tail -5 $O/validation.txt; exit 1; }
tail -3 $O/validation.txt
timeout -k 10 400 python tools/fuzz_parity.py 2000 120 >> $O/validation.txt 2>&1 || { tail -5 $O/validation.txt; exit 1; }
tail -2 $O/validation.txt
timeout -k 10 600 python tools/long_validation.py 1048576 200 100 >> $O/validation.txt 2>&1 || { tail -5 $O/validation.txt; exit 1; }
tail -2 $O/validation.txt
timeout -k 10 300 python tools/fuzz_api.py 3000 60 >> $O/validation.txt 2>&1 || { tail -5 $O/validation.txt; exit 1; }
timeout -k 10 300 python tools/fuzz_slabs.py 3000 12 >> $O/validation.txt 2>&1 || { tail -5 $O/validation.txt; exit 1; }
tail -2 $O/validation.txt
tail -4 $O/validation.txt
