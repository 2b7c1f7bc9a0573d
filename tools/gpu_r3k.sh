#!/bin/bash
export TMPDIR=/tmp
T=gpurun_out
timeout 120 python tools/soak.py 400 $T/soak_r03.json > $T/soak_r03.log 2>&1; tail -1 $T/soak_r03.log | cut -c1-500
for B in 8 32 128; do
  timeout 300 python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch $B: %.1f img/s %.3f ms/step fwd %.1f TF (executed %.1f)' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['executed']['achieved']))" | tee -a $T/batch_sweep_r03.txt
done
