#!/bin/bash
# One GPU-box visit, evidence first: bench line, rocprofv3 kernel stats, HBM-traffic PMC passes, smoke, then the gpu tests.
# Usage: tools/gpu_round.sh [tag] [pytest seconds]
TAG=${1:-r03}
PYT=${2:-1100}
mkdir -p gpurun_out gpurun_out/pmc_traffic_$TAG
export TMPDIR=/tmp
export SSP_TUNE_CACHE=$(pwd)/gpurun_out/tune_cache_$TAG.json
rm -f $SSP_TUNE_CACHE
REPO=$(pwd)
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
# (profiled runs: tuned choices come from the bench run's cache, already verified there - no verify launches in the traces)
export SSP_TUNE_VERIFY=0
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o prof -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extras --profile-run > $REPO/gpurun_out/prof_$TAG.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_traffic_$TAG/$C -o p -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-extras --profile-run > $REPO/gpurun_out/pmc_traffic_$TAG/$C.log 2>&1
done
cd $REPO
unset SSP_TUNE_VERIFY
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" gpurun_out/kernel_stats_$TAG.csv && head -24 "$F" | cut -c1-170
python tools/timeline.py $(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1) > gpurun_out/timeline_$TAG.txt 2>&1
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +8M -delete
python tools/traffic_summary.py gpurun_out/pmc_traffic_$TAG gpurun_out/traffic_$TAG.json | head -40
find gpurun_out/pmc_traffic_$TAG -name "*.csv" -delete
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
tail -2 gpurun_out/smoke_$TAG.log
timeout $PYT python -m pytest tests -m gpu -q -rfP -p no:cacheprovider --durations=15 > gpurun_out/pytest_$TAG.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_$TAG.log | tail -2
grep -E "^(FAILED|ERROR)|passed|failed|yolo-pose|error" gpurun_out/pytest_$TAG.log | tail -40
