# visit 4: first-layer wgrad fp64 finalize, multi drop-in, plan sync, multi-seed un-frozen test, multi-scale sweep
bash tools/gpu_round.sh r05d kernels
PYTEST_ARGS="tests/test_gpu_dropin.py tests/test_gpu_dist.py tests/test_gpu_darknet.py -k multi_py_runs_or_sync_or_full_train_matches_or_two_ranks_model" PYTEST_SECONDS=900 bash tools/gpu_round.sh r05d tests
bash tools/gpu_round.sh r05d benchq
MULTISCALE_ARGS="all 8" bash tools/gpu_round.sh r05d multiscale
