#!/bin/bash
export TMPDIR=/tmp
T=gpurun_out
timeout 300 python -m pytest tests/test_gpu_wino.py -q -p no:cacheprovider 2>&1 | tail -2
rm -f /tmp/c*.json
bash tools/gpu_ab.sh r3i "SSP_TUNE_CACHE=/tmp/c1.json" "SSP_WINO_SHARE_V=0 SSP_TUNE_CACHE=/tmp/c1.json"
timeout 300 python tools/infer_bench.py 2>/dev/null | cut -c1-300
