#!/bin/bash
# Round 4, visit G: kernel timeline of a training step with the gradient reducer active on one rank (why 7 ms slower?)
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
export SSP_TUNE_CACHE=$REPO/gpurun_out/tune_cache_r4g.json
rm -f $SSP_TUNE_CACHE
timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-verify > /dev/null 2>&1
export SSP_TUNE_VERIFY=0
cd /tmp
for M in plain reducer; do
  if [ $M = reducer ]; then export SSP_BENCH_FORCE_REDUCER=1 SSP_BENCH_SKIP_GRAD_CHECK=1; fi
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_r4g_$M -o prof -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extras --profile-run > $REPO/gpurun_out/prof_r4g_$M.log 2>&1
  python $REPO/tools/timeline.py $(find $REPO/gpurun_out/prof_r4g_$M -name "*kernel_trace.csv" | head -1) v > $REPO/gpurun_out/timeline_r4g_$M.txt 2>&1
  head -36 $REPO/gpurun_out/timeline_r4g_$M.txt
done
find $REPO/gpurun_out/prof_r4g_plain $REPO/gpurun_out/prof_r4g_reducer -name "*.csv" -delete
