#!/bin/bash
# Round 4, visit B: the tuner with F(4x4) candidates - bench line (verified against the oracle before timing) + full-size parity tests
mkdir -p gpurun_out
export TMPDIR=/tmp
export SSP_TUNE_CACHE=$(pwd)/gpurun_out/tune_cache_r4b.json
rm -f $SSP_TUNE_CACHE
timeout 1500 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/bench_r4b.json 2> gpurun_out/bench_r4b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r4b.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','verified','step_conv_effective_flop_frac_of_peak','step_mfma_executed_frac_of_peak','kernel_ms_per_step'):
    print(k, d.get(k))
for k in ('roofline','roofline_dgrad','roofline_wgrad','roofline_bwd','roofline_wino_transforms'):
    r=d[k]; print(k, {kk: r[kk] for kk in r if kk in ('achieved','frac','ms_per_step','effective','gemm_kernels_only','winograd_layers','by_family')})
print('verify', {k: v for k, v in (d.get('verify') or {}).items() if not isinstance(v, dict)})
PY
tail -5 gpurun_out/bench_r4b.err
python tools/show_plans.py 64 416 > gpurun_out/plans_r4b.txt 2>&1; cat gpurun_out/plans_r4b.txt | grep layer
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -rfP -p no:cacheprovider > gpurun_out/r4b_fullsize.log 2>&1
grep -E "passed|failed|error|Error" gpurun_out/r4b_fullsize.log | tail -5
