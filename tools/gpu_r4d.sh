#!/bin/bash
# Round 4, visit D: training-step hipGraphs (test + batch sweep with graphs on / off), inference numbers, RCCL rehearsal
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_darknet.py -q -x -rfP -p no:cacheprovider -k "step_graph or train_step or graph_inference" > gpurun_out/r4d_graph_tests.log 2>&1
grep -E "passed|failed|^E  |Warning|warn" gpurun_out/r4d_graph_tests.log | tail -12
for G in 0 1; do
  for B in 8 16; do
    SSP_STEP_GRAPH=$G timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-verify --timers none 2> gpurun_out/r4d_b${B}_g$G.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph=$G batch=$B', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done
timeout 300 python tools/infer_bench.py > gpurun_out/r4d_infer.json 2> gpurun_out/r4d_infer.err; python -c "
import json; d=json.load(open('gpurun_out/r4d_infer.json')); print({k: v for k, v in d.items() if 'eval' in k or 'region' in k.lower()})"
SSP_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-verify > gpurun_out/r4d_rccl_rehearsal.json 2> gpurun_out/r4d_rccl.err; python -c "
import json; d=json.loads(open('gpurun_out/r4d_rccl_rehearsal.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('comm')))"
tail -3 gpurun_out/r4d_rccl.err
