mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$(pwd)
export SSP_TUNE_CACHE=$REPO/gpurun_out/tune_cache_infer5.json
python tools/infer_trace.py 1 6 > /dev/null 2>&1
cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_infer -o p -- python $REPO/tools/infer_trace.py 1 8 > $REPO/gpurun_out/pmc_infer.log 2>&1
cd $REPO
python tools/infer_pmc.py gpurun_out/pmc_infer > gpurun_out/infer_pmc_b1.txt 2>&1; cat gpurun_out/infer_pmc_b1.txt
