#!/bin/bash
# A/B the training step under environment switches on ONE box: tools/gpu_ab.sh TAG "ENV=.. ENV2=.." "..." ; "-" = defaults.
TAG=$1; shift
mkdir -p gpurun_out
export SSP_TUNE_CACHE=$(pwd)/gpurun_out/tune_cache_$TAG.json
for round in 1 2; do
  for o in "$@"; do
    if [ "$o" = "-" ]; then e=""; else e="$o"; fi
    env $e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-extras 2>gpurun_out/ab_$TAG.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-34s %8.2f img/s %7.3f ms  fwd %.1f TF bwd %.1f TF' % ('$o', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_bwd']['achieved']), d['kernel_ms_per_step'])" | tee -a gpurun_out/ab_$TAG.log
  done
done
