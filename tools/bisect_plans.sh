#!/bin/bash
# Find an autotuned plan set that makes tools/step_check_cli.py fail, then re-run THAT set under A/B switches.
mkdir -p gpurun_out
bad=""
for k in 1 2 3 4; do
  f=gpurun_out/tc_bisect_$k.json; rm -f $f
  SSP_FIRST_FUSED=0 SSP_TUNE_CACHE=$(pwd)/$f python tools/step_check_cli.py > gpurun_out/bisect_$k.log 2>&1
  grep -A1 STEPCHECK gpurun_out/bisect_$k.log
  if grep -q "'24.weight', 0.0006" gpurun_out/bisect_$k.log; then bad=$f; break; fi
done
[ -z "$bad" ] && { echo "no failing plan set found"; exit 0; }
echo "=== failing set: $bad"; grep "plans:" gpurun_out/bisect_$k.log
for cfg in "--opt igemm_variant=63" "--opt wgrad_variant=11" "ENV:SSP_BN_FUSE=0" "--opt igemm_xcd=0"; do
  if [[ "$cfg" == ENV:* ]]; then e="${cfg#ENV:}"; a=""; else e="X=1"; a="$cfg"; fi
  echo "--- $cfg"
  env $e SSP_FIRST_FUSED=0 SSP_TUNE_CACHE=$(pwd)/$bad python tools/step_check_cli.py $a 2>&1 | grep -A1 STEPCHECK
done
