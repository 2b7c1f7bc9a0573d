#!/bin/bash
# Round 4, visit N: tail schedule of the backward pass (layer 2's filter gradient released behind its data gradient, the first
# block's filter gradient on the main stream next to it) - A/B on the driver's command, and through the single-rank reducer.
mkdir -p gpurun_out
export TMPDIR=/tmp
A="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
pr='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], "images/s", d["ms_per_step"], "ms/step verified", d.get("verified"), (d.get("verify") or {}).get("grad"), (d.get("comm") or {}).get("exposed_tail_ms"))'
{
timeout 600 python bench.py $A 2>/dev/null | python -c "$pr" tail_sched_on_verified
SSP_TAIL_SCHED=0 timeout 600 python bench.py $A --no-verify 2>/dev/null | python -c "$pr" tail_sched_off
timeout 600 python bench.py $A --no-verify 2>/dev/null | python -c "$pr" tail_sched_on
SSP_TAIL_SCHED=0 timeout 600 python bench.py $A --no-verify 2>/dev/null | python -c "$pr" tail_sched_off
SSP_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py $A --no-verify 2>/dev/null | python -c "$pr" tail_sched_on_single_rank_reducer
} | tee gpurun_out/r4n_tail_sched.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/prof_r4n -o prof -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extras --profile-run > $OLDPWD/gpurun_out/prof_r4n.log 2>&1
cd $OLDPWD
python tools/timeline.py $(find gpurun_out/prof_r4n -name "*kernel_trace.csv" | head -1) v > gpurun_out/timeline_r4n.txt 2>&1
head -3 gpurun_out/timeline_r4n.txt; tail -12 gpurun_out/timeline_r4n.txt | cut -c1-110
find gpurun_out/prof_r4n -name "*kernel_trace.csv" -delete
