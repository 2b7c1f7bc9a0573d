#!/bin/bash
# GPU-box visit without the profiler passes: gpu tests, smoke, bench line.  Usage: tools/gpu_quick.sh [tag] [pytest args...]
TAG=${1:-q}
shift
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -rfP -p no:cacheprovider --durations=15 "$@" > gpurun_out/pytest_$TAG.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|yolo-pose|^E  " gpurun_out/pytest_$TAG.log | tail -60
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
tail -2 gpurun_out/smoke_$TAG.log
timeout 1200 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cat gpurun_out/bench_$TAG.json
tail -5 gpurun_out/bench_$TAG.err
