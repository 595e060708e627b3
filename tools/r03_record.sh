#!/bin/bash
# Round-3 record run on the final tree: bench lines of every BASELINE config + rocprofv3 kernel stats of the default command, written
# under gpurun_out/r03/ (the summaries quoted in DESIGN / README are copied to profiles/r03_*).
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03/rec
mkdir -p $O
cd $R
python bench.py --config 2 --new-tokens 20 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_7b.json 2> $O/bench_7b.err || exit 1
python bench.py --config 4 --steps 12 --warmup 3 --no-cpu-baseline > $O/dense_prefill_7b_bf16.json 2> $O/dense_bf16.err || exit 1
python bench.py --config 4 --vit-fp8 --steps 12 --warmup 3 --no-cpu-baseline > $O/dense_prefill_7b_fp8vit.json 2> $O/dense_fp8.err || exit 1
python bench.py --steps 3600 --warmup 10 --no-cpu-baseline --no-roofline --no-extra-values > $O/bench_1h.json 2> $O/bench_1h.err || exit 1
python bench.py --config 2 --new-tokens 20 --steps 3600 --warmup 10 --no-cpu-baseline --no-roofline --no-extra-values > $O/bench_7b_1h.json 2> $O/bench_7b_1h.err || exit 1
python bench.py --decode-tail --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-values > $O/bench_decode_tail.json 2> $O/bench_decode_tail.err
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --no-cpu-baseline --no-extra-values > $O/prof_bench.json 2> $O/prof_bench.err
find $O/prof -name "*kernel_stats.csv" | head
