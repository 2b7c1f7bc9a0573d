#!/bin/bash
# One GPU-box visit, in stages.  Every record under profiles/ is reproduced by one invocation of this script:
#
#   tools/gpu_round.sh TAG STAGE [STAGE ...]          (gpurun -- 'tools/gpu_round.sh r05 kernels convbench benchq')
#
# Stages (outputs go to gpurun_out/, named *_TAG.*; copy what is to be judged into profiles/):
#   kernels      the per-kernel parity tests (conv / Winograd / first block / head) - the quick check after a kernel change
#   tests        the whole `pytest -m gpu` suite              (PYTEST_PATHS="tests/test_x.py::test_y ..." narrows it, PYTEST_ARGS adds
#                flags, PYTEST_SECONDS bounds it)
#   smoke        __graft_entry__.smoke()
#   bench        the driver's command: python bench.py --steps 20 --warmup 5       (BENCH_ARGS adds flags)
#   benchq       the same without extras / CPU baseline (verify stays on)          (BENCH_ARGS adds flags)
#   ab           whole-step A/B on this box, two interleaved rounds: AB_CFGS="-;@acc_chunk=0;SSP_WINOGRAD=0"  (BENCH_ARGS adds flags)
#                (a cfg is "-" = defaults, VAR=value environment settings and / or @name=value,... = bench.py --opt)
#   profile      rocprofv3 --kernel-trace --stats of 1 warm-up + 3 real steps, per-queue timeline (+ launch list) of the last
#   traffic      rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (own runs, kernel trace only) -> traffic_TAG.json
#   pmcconv      MFMA / LDS counters of single conv launches: PMC_CASES=l9,l18 PMC_OPS=fwd,dgrad PMC_PLANS=0,8006413
#   convbench    tools/conv_bench.py $CONVBENCH_ARGS  (single-layer timings / errors through the C ABI)
#   multiscale   tools/multiscale_check.py $MULTISCALE_ARGS   (default: all 8)
#   soak         tools/soak.py $SOAK_ARGS   (default: 1000 steps -> gpurun_out/soak_TAG.json)
#   run          $RUN_CMD (anything else, logged to run_TAG.log)
# The bench stages write the timed plan choices to gpurun_out/tune_cache_TAG.json; the profiled stages reuse them, so their
# traces hold training steps only.
TAG=${1:?usage: tools/gpu_round.sh TAG STAGE...}
shift
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
export SSP_TUNE_CACHE=$REPO/gpurun_out/tune_cache_$TAG.json
OUT=$REPO/gpurun_out

bench_line() {
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print('no bench line in', sys.argv[1], e); sys.exit(0)
v = d.get('verify') or {}
print('bench: %.1f images/s %.3f ms/step verified=%s roofline.frac=%s launch_units=%s' % (
    d['value'], d['ms_per_step'], d.get('verified'), d['roofline'].get('frac'), (d['roofline'].get('launch_units') or {}).get('frac')))
print('verify:', {k: v.get(k) for k in ('head', 'head64', 'head64_ref', 'conv', 'grad_other_params', 'grad_first_filter', 'worst_grad_params', 'margin')})
print('kernel_ms_per_step:', d.get('kernel_ms_per_step'))
hb = d.get('head_budget')
if hb and not hb.get('pinned'):
    print('head_budget: budget %s chosen deviation %s moved %s cost %.3f ms' % (hb['budget'], hb['head_deviation_of_the_chosen_plans'], hb['moved'], hb['cost_ms_per_forward']))
    for l, rows in sorted(hb['per_layer_candidates'].items(), key=lambda kv: int(kv[0])):
        print('   layer %2s: %s' % (l, ['%s %s ms d=%.2e' % (r['family'], r['ms'], r['head_deviation']) for r in rows]))
for k in ('train_416_b8', 'eval_672_b1', 'multi_cfg_train_step'):
    if k in (d.get('extra') or {}):
        print(k, d['extra'][k])
PY
}

for STAGE in "$@"; do
  echo "=== stage $STAGE ($TAG) ==="
  case "$STAGE" in
    kernels)
      timeout ${PYTEST_SECONDS:-900} python -m pytest tests/test_gpu_kernels.py tests/test_gpu_wino.py tests/test_gpu_first.py tests/test_gpu_head.py \
        -m gpu -q -x -rf -p no:cacheprovider $PYTEST_ARGS > $OUT/pytest_kernels_$TAG.log 2>&1
      grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/pytest_kernels_$TAG.log | tail -30 ;;
    tests)
      timeout ${PYTEST_SECONDS:-1500} python -m pytest ${PYTEST_PATHS:-tests} -m gpu -q -rfs -p no:cacheprovider --durations=15 $PYTEST_ARGS > $OUT/pytest_$TAG.log 2>&1
      grep -E "^(FAILED|ERROR|SKIPPED)|passed|failed|yolo-pose|^E  " $OUT/pytest_$TAG.log | tail -60 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1
      tail -2 $OUT/smoke_$TAG.log ;;
    bench)
      timeout 1500 python bench.py --steps 20 --warmup 5 $BENCH_ARGS > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
      tail -3 $OUT/bench_$TAG.err; bench_line $OUT/bench_$TAG.json ;;
    benchq)
      timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline $BENCH_ARGS > $OUT/benchq_$TAG.json 2> $OUT/benchq_$TAG.err
      tail -3 $OUT/benchq_$TAG.err; bench_line $OUT/benchq_$TAG.json ;;
    ab)
      IFS=';' read -ra CFGS <<< "${AB_CFGS:--}"
      for round in 1 2; do
        for o in "${CFGS[@]}"; do
          e=""; a=""
          for tok in $o; do
            case "$tok" in -) ;; @*) a="--opt ${tok#@}" ;; *) e="$e $tok" ;; esac
          done
          env $e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-extras $BENCH_ARGS $a 2>$OUT/ab_$TAG.err | tail -1 | \
            python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %8.2f img/s %7.3f ms  fwd %.1f TF bwd %.1f TF' % ('$o', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_bwd']['achieved']), d['kernel_ms_per_step'])" | tee -a $OUT/ab_$TAG.log
        done
      done ;;
    profile)
      export SSP_TUNE_VERIFY=0      # choices come from the bench stage's cache, verified there: no verify launches in the trace
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o prof -- \
        python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extras --profile-run $BENCH_ARGS > $OUT/prof_$TAG.log 2>&1)
      unset SSP_TUNE_VERIFY
      F=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
      [ -n "$F" ] && cp "$F" $OUT/kernel_stats_$TAG.csv && head -24 "$F" | cut -c1-170
      T=$(find $OUT/prof_$TAG -name "*kernel_trace.csv" | head -1)
      [ -n "$T" ] && python tools/timeline.py "$T" > $OUT/timeline_$TAG.txt 2>&1 && python tools/timeline.py "$T" verbose > $OUT/timeline_launches_$TAG.txt 2>&1
      head -3 $OUT/timeline_$TAG.txt
      find $OUT/prof_$TAG -name "*kernel_trace.csv" -size +8M -delete ;;
    traffic)
      export SSP_TUNE_VERIFY=0
      mkdir -p $OUT/pmc_traffic_$TAG
      for C in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_traffic_$TAG/$C -o p -- \
          python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-extras --profile-run $BENCH_ARGS > $OUT/pmc_traffic_$TAG/$C.log 2>&1)
      done
      unset SSP_TUNE_VERIFY
      python tools/traffic_summary.py $OUT/pmc_traffic_$TAG $OUT/traffic_$TAG.json | head -40
      find $OUT/pmc_traffic_$TAG -name "*.csv" -delete ;;
    pmcconv)
      bash tools/pmc_conv.sh $TAG "${PMC_CASES:-l9,l18}" "${PMC_OPS:-fwd,dgrad}" "" "${PMC_PLANS:-0}" > $OUT/pmc_conv_$TAG.txt 2>&1
      tail -40 $OUT/pmc_conv_$TAG.txt ;;
    convbench)
      timeout 900 python tools/conv_bench.py $CONVBENCH_ARGS > $OUT/convbench_$TAG.txt 2>&1
      cat $OUT/convbench_$TAG.txt | tail -80 ;;
    multiscale)
      timeout 1500 python tools/multiscale_check.py ${MULTISCALE_ARGS:-all 8} > $OUT/multiscale_$TAG.txt 2>&1
      tail -30 $OUT/multiscale_$TAG.txt ;;
    soak)
      timeout 1500 python tools/soak.py ${SOAK_ARGS:-1000 $OUT/soak_$TAG.json} > $OUT/soak_$TAG.log 2> $OUT/soak_$TAG.err
      head -c 600 $OUT/soak_$TAG.log ;;
    run)
      timeout ${RUN_SECONDS:-900} bash -c "$RUN_CMD" > $OUT/run_$TAG.log 2>&1
      tail -40 $OUT/run_$TAG.log ;;
    *) echo "unknown stage $STAGE" ;;
  esac
done
