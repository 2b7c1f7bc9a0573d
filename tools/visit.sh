# final checks at HEAD: whole gpu suite (+ printed statistics), smoke, RCCL single-rank rehearsal, 1000-step soak
PYTEST_ARGS="-rP" PYTEST_SECONDS=1500 bash tools/gpu_round.sh r05z tests smoke
grep -E "whole-network gradients|head vs float64|vs float64 frozen|yolo-pose B=64 416:" gpurun_out/pytest_r05z.log | cut -c1-600
grep -E "passed|failed" gpurun_out/pytest_r05z.log | tail -2
SSP_BENCH_FORCE_REDUCER=1 BENCH_ARGS="--no-verify" bash tools/gpu_round.sh r05rccl benchq
python -c "
import json; d=json.loads(open('gpurun_out/benchq_r05rccl.json').read().strip().splitlines()[-1]); print('rccl rehearsal', d['value'], d['ms_per_step'], json.dumps(d.get('comm'))[:900])"
SOAK_ARGS="1000 gpurun_out/soak_r05.json" bash tools/gpu_round.sh r05 soak
