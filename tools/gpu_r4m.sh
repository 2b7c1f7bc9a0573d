#!/bin/bash
# Round 4, visit M: evidence set at HEAD (mosaic tiling in): bench line, kernel stats, timeline, HBM traffic, PMC of layers 18 / 23.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_dropin.py -q -x -p no:cacheprovider -k "image_module" 2>&1 | tail -15 | tee gpurun_out/r4m_image_test.log
SSP_EVIDENCE_SKIP_DIRECT_PMC=1 SSP_EVIDENCE_WINO_CASES=l18,l23 bash tools/gpu_r4e.sh r04m
