#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_final_r03.log 2>&1; tail -3 gpurun_out/pytest_final_r03.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default_r03.json 2> gpurun_out/bench_default_r03.err; python -c "
import json; d=json.load(open('gpurun_out/bench_default_r03.json')); print(d['value'], d['ms_per_step'], d['verified'], d['roofline']['traffic'], d['roofline']['frac'], d['roofline']['executed']['frac'], d['cpu_baseline']['value'])"
