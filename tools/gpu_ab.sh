#!/bin/bash
# A/B the training step on ONE box: tools/gpu_ab.sh TAG "cfg" "cfg" ...; a cfg is "-" (defaults), "VAR=1 VAR2=2" (environment)
# and/or "@name=value,..." (bench.py --opt, e.g. "@igemm_variant=63"); two interleaved rounds.
TAG=$1; shift
mkdir -p gpurun_out
export SSP_TUNE_CACHE=$(pwd)/gpurun_out/tune_cache_$TAG.json
for round in 1 2; do
  for o in "$@"; do
    e=""; a=""
    for tok in $o; do
      case "$tok" in
        -) ;;
        @*) a="--opt ${tok#@}" ;;
        *) e="$e $tok" ;;
      esac
    done
    env $e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-extras $a 2>gpurun_out/ab_$TAG.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-34s %8.2f img/s %7.3f ms  fwd %.1f TF bwd %.1f TF' % ('$o', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_bwd']['achieved']), d['kernel_ms_per_step'])" | tee -a gpurun_out/ab_$TAG.log
  done
done
