#!/bin/bash
# round-2 evidence run on the final code: long validation, kernel stats, SQ counters (2D strict), HBM traffic, bench lines
set -o pipefail
O=gpurun_out/r02k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python -c "import __graft_entry__ as g; g.build_product(); g.build_checker()" || exit 1
rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python3 bench.py --no-build --no-alt --no-cpu-baseline > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
echo "stats done"
bash tools/gpu/pmc3.sh r02k/c2d 2d 10 20 strict || exit 1
bash tools/gpu/pmc3.sh r02k/c3d 3d 10 20 strict || exit 1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 bench.py --no-build --steps 10 --warmup 10 --no-alt --no-cpu-baseline > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 bench.py --no-build --steps 10 --warmup 10 --no-alt --no-cpu-baseline > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
echo "pmc done"
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python bench.py --steps 20 --warmup 5 --no-alt > $O/bench_driver_window.json 2> $O/bench2.err || { tail -20 $O/bench2.err; exit 1; }
head -c 1200 $O/bench.json
