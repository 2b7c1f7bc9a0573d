mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
export SSP_TUNE_CACHE=$REPO/gpurun_out/tune_cache_infer.json
python tools/infer_trace.py 1 6 > /dev/null 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_infer -o p -- python $REPO/tools/infer_trace.py 1 12 > $REPO/gpurun_out/prof_infer.log 2>&1
cd $REPO
python tools/infer_trace.py --print $(find gpurun_out/prof_infer -name "*kernel_trace.csv" | head -1) > gpurun_out/infer_trace_b1.txt
cat gpurun_out/infer_trace_b1.txt
grep -c . gpurun_out/tune_cache_infer.json
