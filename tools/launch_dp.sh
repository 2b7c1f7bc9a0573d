#!/bin/bash
# Data-parallel launch on ONE node: one process per GPU over RCCL / xGMI.
#
#   tools/launch_dp.sh N [script [args...]]        default script: bench.py --gpus N --steps 20 --warmup 5
#   tools/launch_dp.sh 8 bench.py --gpus 8 --steps 20 --warmup 5
#   tools/launch_dp.sh 8 /path/to/your_train.py ...          (a script that calls singleshotpose_amd.dist.init_distributed())
#
# What it sets and why:
#   GPU_MAX_HW_QUEUES=8         HIP multiplexes a process's streams over 4 hardware queues by default; RCCL and the process
#                               group bring streams of their own, and with 4 queues the step's two compute streams end up
#                               sharing one (measured on the single-rank rehearsal: 35.7 ms per step instead of 28.8,
#                               profiles/r04_rccl_queues.txt).  Must be in the environment BEFORE the HIP runtime starts:
#                               exporting it here does not depend on the import order inside the script.
#   HSA_ENABLE_IPC_MODE_LEGACY=0  this pool's driver supports dmabuf IPC only (RCCL's intra-node transport)
#   --master-addr 127.0.0.1     the container hostname may not resolve
N=${1:?usage: tools/launch_dp.sh N [script args...]}
shift
HERE=$(cd "$(dirname "$0")/.." && pwd)
if [ $# -eq 0 ]; then set -- "$HERE/bench.py" --gpus "$N" --steps 20 --warmup 5; fi
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
PORT=${MASTER_PORT:-$(python - <<'PY'
import socket
s = socket.socket(); s.bind(('127.0.0.1', 0)); print(s.getsockname()[1]); s.close()
PY
)}
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" "$@"
