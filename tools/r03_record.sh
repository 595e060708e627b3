#!/bin/bash
# Round-3 record run: bench lines + rocprofv3 kernel stats of the default command, written under gpurun_out/r03/ (copied to profiles/).
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python bench.py --config 2 --new-tokens 20 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_7b.json 2> $O/bench_7b.err || exit 1
python bench.py --decode-tail --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-values > $O/bench_decode_tail.json 2> $O/bench_decode_tail.err
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --no-cpu-baseline --no-extra-values > $O/prof_bench.json 2> $O/prof_bench.err
find $O/prof -name "*kernel_stats.csv" | head
