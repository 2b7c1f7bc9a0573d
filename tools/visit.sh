# visit 2: head-error budget of the forward plans
bash tools/gpu_round.sh r05b kernels
BENCH_ARGS="--verify-exact" bash tools/gpu_round.sh r05b benchq
SSP_HEAD_ERR_BUDGET=0 BENCH_ARGS="--verify-exact" bash tools/gpu_round.sh r05b0 benchq
SSP_HEAD_ERR_BUDGET=2.5e-5 bash tools/gpu_round.sh r05b25 benchq
SSP_WINOGRAD=0 bash tools/gpu_round.sh r05bd benchq
