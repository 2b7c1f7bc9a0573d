#!/bin/bash
# Round-3 evidence visit, second pass: same as gpu_r3f's first half with the tune cache kept intact for the profiled runs.
export TMPDIR=/tmp
T=gpurun_out
(nproc; lscpu | grep -E "Model name|Socket|Thread|Core"; echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null || cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null)"; rocminfo | grep -E "Marketing|Compute Unit|Max Clock" | tail -3) > $T/host_r03.txt 2>&1
bash tools/gpu_round.sh r03 1500
