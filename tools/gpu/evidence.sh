#!/bin/bash
# usage: evidence.sh <tag>   -> gpurun_out/<tag>/: the round's evidence runs (bench lines, kernel stats, HBM traffic passes, SQ counter passes)
set -o pipefail
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --no-build > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench default done"
python3 bench.py --no-build --steps 20 --warmup 5 --no-alt > $O/bench_driver_window_5_20.json 2>> $O/bench.err || exit 1
echo "bench driver window done"
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python3 bench.py --no-build --no-alt --no-cpu-baseline > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 bench.py --no-build --steps 10 --warmup 10 --no-alt --no-cpu-baseline > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 bench.py --no-build --steps 10 --warmup 10 --no-alt --no-cpu-baseline > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
echo "traffic passes done"
bash tools/gpu/pmc3.sh $TAG/c2d 2d 10 20 strict || exit 1
echo "all done"
