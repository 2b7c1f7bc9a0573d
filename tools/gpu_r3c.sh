#!/bin/bash
# Round-3 GPU visit C: Winograd filter gradient (parity + timing), label-upload fix check, step A/B, bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino.py -q -rP -p no:cacheprovider > $T/pytest_wino_r3c.log 2>&1
grep -E "winograd |passed|failed|^E  |Error" $T/pytest_wino_r3c.log | cut -c1-300 | tail -30
timeout 400 python tools/conv_bench.py --cases l12,l18,l23,l29 --ops wgrad,wgradw --iters 12 > $T/convbench_wgradw_r3c.txt 2>&1
grep -v amdgpu.ids $T/convbench_wgradw_r3c.txt
timeout 300 python tools/label_upload_probe.py > $T/label_upload_r3c.json 2> $T/label_upload_r3c.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/label_upload_r3c.json'))
for k, v in d.items():
    print(k, [(r['call_us']['median'], r['call_us']['p99'], r['call_us']['max'], r['worst_call']) for r in v][:3], [r['slow_calls'][:4] for r in v][:3])
PY
timeout 200 python -m pytest tests/test_gpu_head.py -q -rP -p no:cacheprovider -k "host_label" 2>&1 | grep -E "RegionLoss host|passed|failed"
export SSP_TUNE_CACHE=$(pwd)/gpurun_out/tune_cache_r3c.json
rm -f $SSP_TUNE_CACHE
bash tools/gpu_ab.sh r3c "-" "SSP_WINOGRAD=0"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -rfP -p no:cacheprovider -k "headline or multi_object or 832 or 224" > $T/pytest_full_r3c.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|yolo-pose|^E  " $T/pytest_full_r3c.log | cut -c1-700 | tail -20
timeout 1500 python bench.py --steps 20 --warmup 5 > $T/bench_r3c.json 2> $T/bench_r3c.err
cat $T/bench_r3c.json | cut -c1-6000; tail -3 $T/bench_r3c.err
