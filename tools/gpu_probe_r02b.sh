mkdir -p gpurun_out
python tools/conv_bench.py --cases l2,l4,l5,l8,l9,l12,l13,l18,l19,l23,l29,l26,l30 --iters 15 > gpurun_out/convbench_r02b.txt 2>&1
cat gpurun_out/convbench_r02b.txt
python tools/infer_bench.py > gpurun_out/infer_r02b.json 2> gpurun_out/infer_r02b.err; tail -30 gpurun_out/infer_r02b.json; tail -3 gpurun_out/infer_r02b.err
bash tools/pmc_conv.sh r02b l8,l23 fwd,dgrad,wgrad > gpurun_out/pmc_conv_r02b.txt 2>&1; tail -40 gpurun_out/pmc_conv_r02b.txt
(lscpu | head -20; nproc; rocminfo | grep -E "Marketing|Compute Unit|Max Clock" | head -8) > gpurun_out/host_r02b.txt 2>&1
