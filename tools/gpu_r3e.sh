#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f /tmp/c*.json
bash tools/gpu_ab.sh r3e "SSP_TUNE_CACHE=/tmp/c256.json" "SSP_WINO_MIN_CHANNELS=128 SSP_TUNE_CACHE=/tmp/c128.json" "SSP_WINO_MIN_CHANNELS=64 SSP_TUNE_CACHE=/tmp/c64.json"
SSP_WINO_MIN_CHANNELS=64 SSP_TUNE_CACHE=/tmp/c64.json python tools/show_plans.py 2>/dev/null | tail -24
