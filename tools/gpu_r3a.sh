#!/bin/bash
# Round-3 GPU visit A: kernel A/B (wgrad unit order, tile order), new parity tests, probes, soak, full bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
export SSP_TUNE_CACHE=$(pwd)/gpurun_out/tune_cache_r3a.json
rm -f $SSP_TUNE_CACHE
T=gpurun_out
(nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; rocm-smi --showclocks | grep -i sclk) > $T/host_r3a.txt 2>&1
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "wgrad" > $T/pytest_wgrad_r3a.log 2>&1; tail -3 $T/pytest_wgrad_r3a.log
timeout 300 python tools/conv_bench.py --cases l18,l19,l23,l29 --ops wgrad --iters 15 \
  --sweep "wgrad_variant=12;-;wgrad_split=8;wgrad_split=12;wgrad_split=14;wgrad_split=16;wgrad_variant=12;-" > $T/convbench_wgrad_r3a.txt 2>&1
cat $T/convbench_wgrad_r3a.txt | grep -v amdgpu.ids
timeout 300 python tools/conv_bench.py --cases l18,l23,l29 --ops fwd,dgrad --iters 15 --sweep "-;igemm_variant=80;-;igemm_variant=80" > $T/convbench_order_r3a.txt 2>&1
cat $T/convbench_order_r3a.txt | grep -v amdgpu.ids
timeout 300 python tools/conv_bench.py --cases l2,l4,l5,l8,l9,l12,l13,l18,l19,l23,l29,l26,l30 --iters 15 > $T/convbench_r3a.txt 2>&1
cat $T/convbench_r3a.txt | grep -v amdgpu.ids
bash tools/gpu_ab.sh r3a "-" "@wgrad_variant=12" "@igemm_variant=80"
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_head.py tests/test_gpu_dist.py -q -rfP -p no:cacheprovider --durations=10 > $T/pytest_new_r3a.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|yolo-pose|visit|RegionLoss host|^E  " $T/pytest_new_r3a.log | tail -40
timeout 300 python -m pytest tests/test_gpu_darknet.py -q -rf -p no:cacheprovider -k "multiscale or plan_cache" > $T/pytest_ms_r3a.log 2>&1; tail -3 $T/pytest_ms_r3a.log
timeout 200 python tools/label_upload_probe.py > $T/label_upload_r3a.json 2> $T/label_upload_r3a.err; cat $T/label_upload_r3a.json | cut -c1-1500
timeout 200 python tools/soak.py 400 $T/soak_r3a.json > $T/soak_r3a.log 2>&1; tail -2 $T/soak_r3a.log | cut -c1-1200
timeout 1500 python bench.py --steps 20 --warmup 5 > $T/bench_r3a.json 2> $T/bench_r3a.err
cat $T/bench_r3a.json; tail -3 $T/bench_r3a.err
