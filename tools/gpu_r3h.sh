#!/bin/bash
export TMPDIR=/tmp
T=gpurun_out
REPO=$(pwd)
timeout 100 python -m pytest tests/test_gpu_head.py -q -rP -p no:cacheprovider -k host_label 2>&1 | grep -E "RegionLoss host|passed|failed"
timeout 500 python tools/multiscale_check.py all 2 $T/multiscale_all_r03.json > $T/multiscale_all_r03.log 2>&1; grep -E "^size|tune rej" $T/multiscale_all_r03.log | cut -c1-170
cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/infer_trace_r03 -o t -- python $REPO/tools/infer_trace.py 1 12 > $REPO/gpurun_out/infer_trace_r03.log 2>&1
cd $REPO
python tools/infer_trace.py --print $(find gpurun_out/infer_trace_r03 -name "*kernel_trace.csv" | head -1) > $T/infer_trace_b1_r03.txt 2>&1; tail -75 $T/infer_trace_b1_r03.txt | cut -c1-150
find gpurun_out/infer_trace_r03 -name "*.csv" -delete
