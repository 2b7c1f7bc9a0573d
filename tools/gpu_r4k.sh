#!/bin/bash
# Round 4, visit K: 2 x 2 image mosaic tiling of the Winograd launches - parity, per-layer timing, step time.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_fullsize.py tests/test_gpu_darknet.py -q -x -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r4k_tests.log
timeout 600 python tools/conv_bench.py --cases l12,l18,l23,l29 --plans 0,8006413,8012813 --ops fwd,dgrad,wgradw 2>&1 | tee gpurun_out/r4k_convbench.txt
timeout 300 python tools/conv_bench.py --cases l18,l23 --B 8 --plans 0,8006413 --ops fwd,dgrad,wgradw 2>&1 | tee gpurun_out/r4k_convbench_b8.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4k_bench.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4k_bench.json').read())
print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'roofline', d['roofline']['frac'], 'wino', d.get('roofline_wino_transforms', {}).get('ms_per_step'))
print({k: v for k, v in d.get('extra', {}).items() if not isinstance(v, dict)})
for k, v in d.get('extra', {}).items():
    if isinstance(v, dict):
        print(k, {a: b for a, b in v.items() if not isinstance(b, (dict, list))})
PY
