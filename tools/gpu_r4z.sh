#!/bin/bash
# Round 4, final visit: the whole GPU suite, smoke(), and the default bench command at HEAD.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf -p no:cacheprovider > gpurun_out/pytest_r04_final.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_r04_final.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/pytest_r04_final.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke | tail -2 | tee gpurun_out/smoke_r04_final.log
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r04_final.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_r04_final.json').read())
print(d['value'], d['ms_per_step'], d['verified'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_wino_transforms']['traffic'], d['steps'], d['warmup'])
PY
