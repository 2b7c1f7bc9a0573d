#!/bin/bash
# Round 4 evidence visit (no pytest): driver-command bench line, rocprofv3 kernel stats + timeline, HBM-traffic PMC passes,
# PMC counter sets of single launches (layers 2, 4, 8, 9, 23).  Usage: tools/gpu_r4e.sh [tag]
TAG=${1:-r04}
mkdir -p gpurun_out gpurun_out/pmc_traffic_$TAG
export TMPDIR=/tmp
export SSP_TUNE_CACHE=$(pwd)/gpurun_out/tune_cache_$TAG.json
rm -f $SSP_TUNE_CACHE
REPO=$(pwd)
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['verified'], d['roofline']['frac'], d['roofline']['launch_units']['frac'], d['kernel_ms_per_step']); print(json.dumps(d.get('extra'))[:1500]); print(json.dumps(d.get('cpu_baseline'))[:300])"
tail -3 gpurun_out/bench_$TAG.err
export SSP_TUNE_VERIFY=0
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o prof -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extras --profile-run > $REPO/gpurun_out/prof_$TAG.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_traffic_$TAG/$C -o p -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-extras --profile-run > $REPO/gpurun_out/pmc_traffic_$TAG/$C.log 2>&1
done
cd $REPO
unset SSP_TUNE_VERIFY
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" gpurun_out/kernel_stats_$TAG.csv && head -30 "$F" | cut -c1-150
python tools/timeline.py $(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1) > gpurun_out/timeline_$TAG.txt 2>&1
head -50 gpurun_out/timeline_$TAG.txt
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +8M -delete
python tools/traffic_summary.py gpurun_out/pmc_traffic_$TAG gpurun_out/traffic_$TAG.json 2 | tail -5
python -c "
import json; d=json.load(open('gpurun_out/traffic_$TAG.json')); print(d['_meta'])"
find gpurun_out/pmc_traffic_$TAG -name "*.csv" -delete
unset SSP_TUNE_CACHE
if [ -z "$SSP_EVIDENCE_SKIP_DIRECT_PMC" ]; then
bash tools/pmc_conv.sh ${TAG}a l2,l4,l9 fwd,dgrad,wgrad "" 0 > gpurun_out/pmc_conv_${TAG}a.txt 2>&1
fi
bash tools/pmc_conv.sh ${TAG}b ${SSP_EVIDENCE_WINO_CASES:-l4,l8,l23} fwd,dgrad,wgradw "" 8006413 > gpurun_out/pmc_conv_${TAG}b.txt 2>&1
find gpurun_out/pmc_${TAG}a gpurun_out/pmc_${TAG}b -name "*.csv" -delete 2>/dev/null
grep -E "grid=|MFMA pipe busy" gpurun_out/pmc_conv_${TAG}a.txt gpurun_out/pmc_conv_${TAG}b.txt | head -80
