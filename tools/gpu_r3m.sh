#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_darknet.py -q -p no:cacheprovider -k "plan_cache or multiscale" 2>&1 | tail -2
timeout 900 python tools/soak_multiscale.py 30 5 gpurun_out/soak_multiscale_r03.json > gpurun_out/soak_multiscale_r03.log 2>&1
grep -E "visit" gpurun_out/soak_multiscale_r03.log | awk 'NR%3==0' | cut -c1-230; tail -1 gpurun_out/soak_multiscale_r03.log
