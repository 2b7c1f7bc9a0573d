# round-5 evidence visit: driver bench line, kernel statistics + timeline, HBM traffic, PMC of single launches, multi-scale sweep
rm -f gpurun_out/tune_cache_r05.json
(nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; rocm-smi --showclocks 2>/dev/null | grep -i sclk) > gpurun_out/host_r05.txt 2>&1
bash tools/gpu_round.sh r05 bench profile traffic
PMC_CASES=l2,l9,l18 PMC_OPS=fwd,dgrad PMC_PLANS=0,8006413 bash tools/gpu_round.sh r05 pmcconv
MULTISCALE_ARGS="all 8" bash tools/gpu_round.sh r05 multiscale
