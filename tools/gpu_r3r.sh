#!/bin/bash
export TMPDIR=/tmp
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
rm -f /tmp/c1.json
bash tools/gpu_ab.sh r3r "SSP_TUNE_CACHE=/tmp/c1.json" "SSP_SIDE_PRIORITY=-1 SSP_TUNE_CACHE=/tmp/c1.json" "SSP_SIDE_PRIORITY=1 SSP_TUNE_CACHE=/tmp/c1.json"
