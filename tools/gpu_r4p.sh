#!/bin/bash
# Round 4, visit P: full-network parity at the resolutions whose deepest maps tile as 2x2 image mosaics (H = 9, 13, 17, 21, 25),
# batch 8 (two mosaics) and batch 7 (a phantom image in the last mosaic); 1000-step soak of the headline step at HEAD.
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "# tools/multiscale_check.py 288,416,544,672,800 8   (HEAD: mosaic tiling; every shape's deepest 3x3 layers run F(4x4) over 2x2 image mosaics)"
timeout 900 python tools/multiscale_check.py 288,416,544,672,800 8 2>&1 | grep -E "^size|tune rejections|FAIL|Error|error"
echo "# tools/multiscale_check.py 288,416 7   (a phantom image in the second mosaic)"
timeout 600 python tools/multiscale_check.py 288,416 7 2>&1 | grep -E "^size|tune rejections|FAIL|Error|error"
} | tee gpurun_out/r4p_multiscale_parity.txt
timeout 600 python tools/soak.py 1000 gpurun_out/r4p_soak.json 2>&1 | tail -2
