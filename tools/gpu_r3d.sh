#!/bin/bash
# Round-3 GPU visit D: per-kernel breakdown of the Winograd step (rocprofv3 stats + timeline), remaining probes.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=gpurun_out
REPO=$(pwd)
export SSP_TUNE_CACHE=$REPO/gpurun_out/tune_cache_r3d.json
rm -f $SSP_TUNE_CACHE
timeout 300 python -m pytest tests/test_gpu_wino.py tests/test_gpu_head.py -q -rP -p no:cacheprovider -k "wgrad or host_label" 2>&1 | grep -E "winograd |RegionLoss host|passed|failed|^E  " | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $T/bench_r3d.json 2> $T/bench_r3d.err
python -c "
import json; d=json.load(open('gpurun_out/bench_r3d.json')); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r3d -o prof -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extras > $REPO/gpurun_out/prof_r3d.log 2>&1
cd $REPO
F=$(find gpurun_out/prof_r3d -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" gpurun_out/kernel_stats_r3d.csv && head -40 "$F" | cut -c1-150
python tools/timeline.py $(find gpurun_out/prof_r3d -name "*kernel_trace.csv" | head -1) > gpurun_out/timeline_r3d.txt 2>&1
head -60 gpurun_out/timeline_r3d.txt
find gpurun_out/prof_r3d -name "*kernel_trace.csv" -size +8M -delete
timeout 200 python tools/label_upload_probe.py > $T/label_upload_r3d.json 2> $T/label_upload_r3d.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/label_upload_r3d.json'))
for k, v in d.items():
    print(k, [(r['call_us']['median'], r['call_us']['p99'], r['call_us']['max']) for r in v], [len(r['slow_calls']) for r in v])
PY
