#!/bin/bash
# Round 4, visit A: F(4x4,3x3) kernels - parity tests, then per-layer timings F(2x2) vs F(4x4) vs direct.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_wino.py -q -x -rfP -p no:cacheprovider > gpurun_out/r4a_wino_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/r4a_wino_tests.log | tail -3
grep -E "winograd (vs|wgrad)" gpurun_out/r4a_wino_tests.log | head -60
timeout 600 python tools/conv_bench.py --cases l8,l12,l18,l23,l29 --ops fwd,dgrad,wgrad,wgradw --iters 10 --plans 0,9006413,9006414,8006413,8006414,8012813,8012814 > gpurun_out/r4a_convbench.txt 2>&1
cat gpurun_out/r4a_convbench.txt | grep -v amdgpu.ids
timeout 300 python tools/conv_bench.py --cases l4,l6 --ops fwd,wgrad,wgradw --iters 10 --plans 0,8006413,8012813 > gpurun_out/r4a_convbench_l4.txt 2>&1
cat gpurun_out/r4a_convbench_l4.txt | grep -v amdgpu.ids
