#!/bin/bash
# One GPU-box visit: gpu tests, bench line, rocprofv3 kernel stats.  Usage: tools/gpu_check.sh [tag]
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_$TAG.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 3000 gpurun_out/bench_$TAG.err
cat gpurun_out/bench_$TAG.json
REPO=$(pwd)
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o prof -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof_$TAG.log 2>&1
cd $REPO
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -3
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -25 "$F"
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
tail -5 gpurun_out/pytest_$TAG.log
