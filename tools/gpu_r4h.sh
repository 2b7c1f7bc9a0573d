#!/bin/bash
# Round 4, visit H: does the early shared side stream restore the two-stream overlap under a live RCCL group?
mkdir -p gpurun_out
export TMPDIR=/tmp
A="--steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-verify"
pr='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]; print(sys.argv[1], d["value"], "images/s", d["ms_per_step"], "ms/step; fwd units", d["roofline"]["ms_per_step"], "ms", (d.get("comm") or {}).get("bucket_issue_to_done_ms"), (d.get("comm") or {}).get("exposed_tail_ms"))'
{
timeout 600 python bench.py $A 2>/dev/null | python -c "$pr" plain
SSP_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py $A 2>/dev/null | python -c "$pr" reducer_early_side_stream
SSP_BENCH_FORCE_REDUCER=1 GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py $A 2>/dev/null | python -c "$pr" reducer_early_side_stream_8_hw_queues
GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py $A 2>/dev/null | python -c "$pr" plain_8_hw_queues
} | tee gpurun_out/r4h_rccl_queues.txt
timeout 600 python -m pytest tests/test_gpu_darknet.py tests/test_gpu_dist.py -q -x -p no:cacheprovider 2>&1 | tail -3
