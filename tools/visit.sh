# visit 5: multi drop-in tests, plan sync, multi-seed statistic, full-size bars, profile of a step (launch list)
PYTEST_ARGS="tests/test_gpu_dropin.py::test_unmodified_train_multi_py_runs_and_matches_the_cpu_reference tests/test_gpu_dropin.py::test_unmodified_valid_multi_py_runs_and_matches_the_cpu_reference tests/test_gpu_dist.py tests/test_gpu_darknet.py::test_full_train_matches_reference tests/test_gpu_fullsize.py tests/test_gpu_dropin.py::test_rawbatch_device_entry_points_and_float_mode tests/test_gpu_dropin.py::test_dropin_dataset_epoch_is_the_reference_epoch_byte_for_byte" PYTEST_SECONDS=1200 bash tools/gpu_round.sh r05e tests
grep -E "whole-network|head vs float64|passed|failed" gpurun_out/pytest_r05e.log | cut -c1-400
bash tools/gpu_round.sh r05e benchq profile
grep -n "rocclr\|Fill" gpurun_out/timeline_launches_r05e.txt | head -60
