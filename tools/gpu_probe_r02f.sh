mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "conv or candidate" 2>&1 | tail -4
export SSP_TUNE_CACHE=$REPO/gpurun_out/tune_cache_infer4.json
python tools/infer_trace.py 1 6 > /dev/null 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_infer4 -o p -- python $REPO/tools/infer_trace.py 1 12 > $REPO/gpurun_out/prof_infer4.log 2>&1
cd $REPO
python tools/infer_trace.py --print $(find gpurun_out/prof_infer4 -name "*kernel_trace.csv" | head -1) > gpurun_out/infer_trace_b1_v4.txt
grep -v splitk gpurun_out/infer_trace_b1_v4.txt | awk '{print $2, $4, $5, $6, $7}' | grep -E "conv|wall"
cat gpurun_out/tune_cache_infer4.json | tr -d '\n' | cut -c1-1500; echo
unset SSP_TUNE_CACHE
python tools/infer_bench.py 2>/dev/null | cut -c1-240
export SSP_TUNE_CACHE=$REPO/gpurun_out/tune_cache_train4.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-extras 2>/dev/null | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-extras 2>/dev/null | cut -c1-200
python tools/show_plans.py 2>/dev/null | tail -30
