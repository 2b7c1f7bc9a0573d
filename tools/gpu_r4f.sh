#!/bin/bash
# Round 4, visit F: step-graph modes at batch 8 (eager / two-stream graph / one-stream graph), and what the forced single-rank
# reducer costs (plain bench vs reducer vs reducer without the gradient check)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_darknet.py -q -x -p no:cacheprovider -k "step_graph" 2>&1 | tail -2
{
for G in 0 1 2; do
  SSP_STEP_GRAPH=$G timeout 300 python bench.py --batch 8 --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-verify --timers none 2> gpurun_out/r4f_b8_g$G.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SSP_STEP_GRAPH=$G batch 8:', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"
done
} | tee gpurun_out/r4f_step_graph.txt
A="--steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-verify"
pr='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]; print(sys.argv[1], d["value"], d["ms_per_step"], "fwd", d["roofline"]["ms_per_step"], "wino_fwd", k["wino_fwd"], "bn_act", k["bn_act"])'
timeout 600 python bench.py $A 2>/dev/null | python -c "$pr" plain
SSP_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py $A 2>/dev/null | python -c "$pr" reducer
SSP_BENCH_FORCE_REDUCER=1 SSP_BENCH_SKIP_GRAD_CHECK=1 timeout 600 python bench.py $A 2>/dev/null | python -c "$pr" reducer_nocheck
timeout 600 python bench.py $A 2>/dev/null | python -c "$pr" plain_again
