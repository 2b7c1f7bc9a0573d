# visit 3: head-error budget of the forward plans
BENCH_ARGS="--verify-exact" bash tools/gpu_round.sh r05c benchq
SSP_HEAD_ERR_BUDGET=2.5e-5 bash tools/gpu_round.sh r05c25 benchq
SSP_HEAD_ERR_BUDGET=0 bash tools/gpu_round.sh r05c0 benchq
