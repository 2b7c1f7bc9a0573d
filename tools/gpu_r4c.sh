#!/bin/bash
# Round 4, visit C: the whole gpu suite + smoke on HEAD (F(4x4) plans, thin-dgrad Winograd, batched-wgrad tile choice)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -rfP -p no:cacheprovider --durations=12 > gpurun_out/pytest_r4c.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/pytest_r4c.log | tail -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r4c.log 2>&1
tail -2 gpurun_out/smoke_r4c.log
timeout 300 python tools/conv_bench.py --cases l4,l18,l23 --ops dgrad,wgradw --iters 10 --plans 0,8006413 > gpurun_out/r4c_convbench.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4c_convbench.txt
