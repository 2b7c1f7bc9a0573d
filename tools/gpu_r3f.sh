#!/bin/bash
# Round-3 final evidence visit: tools/gpu_round.sh (bench line of the driver's command, rocprofv3 kernel stats, HBM-traffic
# PMC passes, timeline, smoke, the whole gpu test suite) + PMC counter sets of single conv launches + latency pieces.
export TMPDIR=/tmp
T=gpurun_out
bash tools/gpu_round.sh r03 1500
bash tools/pmc_conv.sh r03 l8,l23 fwd,dgrad,wgrad,wgradw "" 0,9006413 > $T/pmc_conv_r03.txt 2>&1; tail -5 $T/pmc_conv_r03.txt
timeout 300 python tools/infer_bench.py > $T/infer_r03.json 2> $T/infer_r03.err; cat $T/infer_r03.json
timeout 200 python tools/label_upload_probe.py > $T/label_upload_r03.json 2> $T/label_upload_r03.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/label_upload_r03.json'))
for k, v in d.items():
    print(k, [(r['call_us']['median'], r['call_us']['p99'], r['call_us']['max']) for r in v], [len(r['slow_calls']) for r in v])
PY
timeout 120 python tools/soak.py 400 $T/soak_r03.json > $T/soak_r03.log 2>&1; tail -1 $T/soak_r03.log | cut -c1-600
timeout 400 python tools/multiscale_check.py 224,320,416,512,608,704,832 2 $T/multiscale_r03.json > $T/multiscale_r03.log 2>&1; tail -9 $T/multiscale_r03.log | cut -c1-200
timeout 100 python tools/show_plans.py > $T/plans_r03.txt 2>/dev/null; tail -24 $T/plans_r03.txt
(nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; rocminfo | grep -E "Marketing|Compute Unit|Max Clock" | head -8) > $T/host_r03.txt 2>&1
