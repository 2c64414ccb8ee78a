#!/bin/bash
# the round-end checks as the driver runs them: every GPU test in one process, then smoke()
set -o pipefail
O=gpurun_out/r02tests; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -14 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
