#!/bin/bash
# Round-3 GPU visit B: Winograd kernels (parity + stand-alone timing), label-upload probe by sub-step, step A/B, bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino.py -q -x -rP -p no:cacheprovider > $T/pytest_wino_r3b.log 2>&1
grep -E "winograd vs|passed|failed|^E  |Error" $T/pytest_wino_r3b.log | tail -30
timeout 400 python tools/conv_bench.py --cases l12,l18,l23,l29 --ops fwd,dgrad,wfilt --iters 12 --plans 0,9006413,9006414,9012813,9012814 > $T/convbench_wino_r3b.txt 2>&1
grep -v amdgpu.ids $T/convbench_wino_r3b.txt
timeout 300 python tools/label_upload_probe.py > $T/label_upload_r3b.json 2> $T/label_upload_r3b.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/label_upload_r3b.json'))
for k, v in d.items():
    print(k, [(r['call_us']['median'], r['call_us']['p99'], r['call_us']['max'], r['worst_call']) for r in v][:2], [r['slow_calls'][:6] for r in v][:2])
PY
export SSP_TUNE_CACHE=$(pwd)/gpurun_out/tune_cache_r3b.json
rm -f $SSP_TUNE_CACHE
bash tools/gpu_ab.sh r3b "-" "SSP_WINOGRAD=0"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -rfP -p no:cacheprovider -k "headline or multi_object or 608 or 352" > $T/pytest_full_r3b.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|yolo-pose|^E  " $T/pytest_full_r3b.log | cut -c1-900 | tail -20
timeout 1500 python bench.py --steps 20 --warmup 5 > $T/bench_r3b.json 2> $T/bench_r3b.err
cat $T/bench_r3b.json; tail -3 $T/bench_r3b.err
