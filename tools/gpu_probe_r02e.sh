mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_darknet.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
export SSP_TUNE_CACHE=$REPO/gpurun_out/tune_cache_infer.json
python tools/infer_trace.py 1 6 > /dev/null 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_infer2 -o p -- python $REPO/tools/infer_trace.py 1 12 > $REPO/gpurun_out/prof_infer2.log 2>&1
cd $REPO
python tools/infer_trace.py --print $(find gpurun_out/prof_infer2 -name "*kernel_trace.csv" | head -1) > gpurun_out/infer_trace_b1_v2.txt
grep -E "splitk|256, 32|128, 64|forward wall" gpurun_out/infer_trace_b1_v2.txt
unset SSP_TUNE_CACHE
python tools/infer_bench.py 2>/dev/null | cut -c1-400
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-extras 2>/dev/null | cut -c1-300
