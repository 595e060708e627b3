#!/bin/bash
# Round-2 record run: bench lines + rocprofv3 kernel stats, written under gpurun_out/r02/ (copied to profiles/ afterwards).
# usage: r02_record.sh [quick]   (quick: skip the two hour-long streams)
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || exit 1
python bench.py --steps 20 --warmup 5 --sampling temperature --no-roofline --no-cpu-baseline > $O/bench_sampling_temperature.json 2> $O/s1.err
python bench.py --steps 20 --warmup 5 --sampling hf-default --no-roofline --no-cpu-baseline > $O/bench_sampling_hf_default.json 2> $O/s2.err
python bench.py --model 7b --fps 2 --window 4096 --new-tokens 20 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_7b.json 2> $O/bench_7b.err
python bench.py --scenario dense_prefill --model 7b --fps 2 --window 4096 --prefill-chunks 300 --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > $O/dense_prefill_7b_bf16.json 2> $O/dp1.err
python bench.py --scenario dense_prefill --model 7b --fps 2 --window 4096 --prefill-chunks 300 --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --vit-fp8 > $O/dense_prefill_7b_fp8vit.json 2> $O/dp2.err
python tools/efficiency_modes.py --model 2b --chunks 240 --out $O/efficiency_modes_2b.json > $O/eff.log 2>&1
if [ "$1" != "quick" ]; then
  python bench.py --steps 3600 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_1h.json 2> $O/bench_1h.err
  python bench.py --model 7b --fps 2 --window 4096 --new-tokens 20 --steps 3600 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_7b_1h.json 2> $O/bench_7b_1h.err
fi
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof_bench.err
find $O/prof -name "*kernel_stats.csv" | head
