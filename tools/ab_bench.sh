#!/bin/bash
# A/B the whole training step on ONE box: tools/ab_bench.sh "<opt A>" "<opt B>" ... (each: name=value,... or "-" for defaults); two rounds, interleaved
for round in 1 2; do
  for o in "$@"; do
    if [ "$o" = "-" ]; then a=""; else a="--opt $o"; fi
    timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s %8.2f img/s %7.3f ms  fwd %.1f TF bwd %.1f TF' % ('$o', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_bwd']['achieved']), d['kernel_ms_per_step'])"
  done
done
