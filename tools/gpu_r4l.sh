#!/bin/bash
# Round 4, visit L: drop-in dataset (GPU augmentation inside the unmodified train.py / DataLoader), plan eviction, multi-scale soak.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_image.py -q -x -p no:cacheprovider -k "dataset or train_py or image" 2>&1 | tail -5 | tee gpurun_out/r4l_dropin_tests.log
timeout 600 python -m pytest tests/test_gpu_darknet.py -q -x -p no:cacheprovider -k "evicted or plan_cache or multiscale" 2>&1 | tail -5 | tee gpurun_out/r4l_plan_tests.log
timeout 600 python tools/soak_multiscale.py 30 4 gpurun_out/r4l_soak_multiscale.json 2>&1 | tail -3
