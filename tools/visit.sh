# visit 7: A/B on one box - early V, budget, chunked accumulation; batch 64 and batch 8
AB_CFGS="-;SSP_WINO_EARLY_V=0;SSP_HEAD_ERR_BUDGET=0" bash tools/gpu_round.sh r05g ab
BENCH_ARGS="--batch 8" AB_CFGS="-;SSP_WINO_EARLY_V=0;SSP_HEAD_ERR_BUDGET=0;@acc_chunk=0" bash tools/gpu_round.sh r05g8 ab
