#!/bin/bash
# Round-6 GPU session runner (through gpurun).  usage: tools/gpu_r6.sh <tag> <deadline_s> <step> [<step> ...]
# Steps not named here are handed to tools/gpu_r5.sh (which hands on to gpu_r4.sh).
set -u
TAG=$1; DEADLINE=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
T0=$(date +%s)
left() { echo $(( DEADLINE - ($(date +%s) - T0) )); }
mkdir -p $OUT /tmp/idx /tmp/prof
cd $R
export TMPDIR=/tmp NANN_TEST_INDEX_CACHE=/tmp/idx
BENCH="python $R/bench.py --index-cache /tmp/idx"
for STEP in "$@"; do
  if [ $(left) -lt 45 ]; then echo "SKIP $STEP (deadline)"; continue; fi
  echo "=== $STEP (left $(left) s)"
  case $STEP in
    shim)  # every kernel of the TF op shim through the functional op-kernel model
      timeout 900 python -m pytest tests/test_tf_shim_gpu.py tests/test_tf_shim.py -q --timeout 600 ${SHIM_ARGS:-} > $OUT/pytest_shim_$TAG.log 2>&1
      tail -5 $OUT/pytest_shim_$TAG.log; grep -E "^(E  |FAILED|ERROR)" $OUT/pytest_shim_$TAG.log | head -40 ;;
    tests_k)  # a selection of the GPU suite: TESTS_K='expr' [TESTS_F='files']
      timeout 1500 python -m pytest ${TESTS_F:-tests} -m gpu -q --timeout 900 -k "${TESTS_K}" > $OUT/pytest_k_$TAG.log 2>&1
      tail -5 $OUT/pytest_k_$TAG.log; grep -E "^(E  |FAILED|ERROR)" $OUT/pytest_k_$TAG.log | head -40 ;;
    tests_all)
      timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/pytest_all_$TAG.log 2>&1
      tail -5 $OUT/pytest_all_$TAG.log; grep -E "^(E  |FAILED|ERROR)" $OUT/pytest_all_$TAG.log | head -40 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
    bench)  # the driver's command
      timeout 600 python bench.py > $OUT/bench_default_$TAG.json 2> $OUT/bench_default_$TAG.err
      python - <<PY
import json
d = json.loads(open('$OUT/bench_default_$TAG.json').read().strip().splitlines()[-1])
print('value', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roofline', json.dumps(d['roofline'])[:600])
PY
      ;;
    replicas_dry)  # bench.py --gpus 2 --replicas with both ranks on the one GPU (gloo carries the barriers): the code path, not a rate
      NANN_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --replicas --dist-backend gloo --items 200000 --batch 1024 --steps 3 --warmup 1 --index-cache /tmp/idx --no-secondary --no-cpu-baseline > $OUT/bench_replicas_dry_$TAG.json 2> $OUT/bench_replicas_dry_$TAG.err
      python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_replicas_dry_$TAG.json').read().strip().splitlines()[-1])
    print('replicas dry: value', d['value'], d['unit'], 'n_gpus', d['n_gpus'], 'parallelism', d['config']['parallelism'][:60], 'parity', d.get('parity'))
except Exception as e:
    print('replicas dry parse failed', e); print(open('$OUT/bench_replicas_dry_$TAG.err').read()[-1500:])
PY
      ;;
    eval_vars)  # phase-repeat builds of the evaluation traversal (a phase run twice costs what it costs once): shipped, then var_ev_*
      for L in $R/nann_amd/_build/libnann_hip.so $R/nann_amd/_build/var_ev_*/libnann_hip.so; do
        V=$(basename $(dirname $L)); [ "$V" = "_build" ] && V=shipped
        NANN_HIP_LIB=$L timeout 300 python tools/eval_bench.py /tmp/idx 1024 2 > $OUT/eval_var_${V}_$TAG.json 2> $OUT/eval_var_${V}_$TAG.err
        python - <<PY
import json
try:
    d = json.loads(open('$OUT/eval_var_${V}_$TAG.json').read().strip().splitlines()[-1])
    print('EVALVAR %-10s defaults %.3f ms  wide %.3f ms  (bit-identical %s/%s)' % ('$V', d['defaults_400_200_100']['ms_per_launch'], d['wide_2000_1000_500']['ms_per_launch'], d['defaults_400_200_100']['bit_identical'], d['wide_2000_1000_500']['bit_identical']))
except Exception as e:
    print('EVALVAR $V failed', e)
PY
      done ;;
    eval_pmc)  # fabric-side bytes of the evaluation traversal (FETCH_SIZE, WRITE_SIZE: one pass each, counters only) -> eval_pmc_$TAG.txt
      # EVAL_SHAPE="4000000 256 bf16 256": another shard shape (tools/eval_bench.py's trailing arguments)
      ( cd /tmp; python $R/tools/eval_bench.py /tmp/idx 1024 0 ${EVAL_SHAPE:-} > /dev/null 2>&1
        for C in FETCH_SIZE WRITE_SIZE; do
          rm -rf /tmp/prof/pe_$C
          timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/prof/pe_$C -o pmc -- python $R/tools/eval_bench.py /tmp/idx 1024 0 ${EVAL_SHAPE:-} > /dev/null 2>&1
          python - <<PY
import csv, glob
vals = []
for f in glob.glob("/tmp/prof/pe_$C/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_search_eval" in r.get("Kernel_Name", ""):
            vals.append(float(r["Counter_Value"]))
print("$C", "k_search_eval dispatches", len(vals), "values", [round(v, 1) for v in vals])
PY
        done ) | tee $OUT/eval_pmc_$TAG.txt ;;
    mlp_batches)  # configs[2] split-f16 at batch 1024 / 2048 / 4096, fused kernel and pipeline of phases (steady state)
      for B in ${MLP_BATCHES:-1024 2048 4096}; do for M in fused phased; do
        if [ $(left) -lt 60 ]; then echo "SKIP $B $M"; continue; fi
        NANN_MLP_FORM=$M timeout 200 $BENCH --scorer mlp --batch $B --steps $(( 150 * 1024 / B )) --warmup $(( 100 * 1024 / B )) --no-secondary --no-cpu-baseline > $OUT/mlp_b${B}_${M}_$TAG.json 2> $OUT/mlp_b${B}_${M}_$TAG.err
        python - <<PY
import json
try:
    d = json.loads(open('$OUT/mlp_b${B}_${M}_$TAG.json').read().strip().splitlines()[-1])
    print('MLP batch %5d %-6s qps %9.0f ms/step %7.4f frac_mfma %s frac_hbm %s' % ($B, '$M', d['value'], d['ms_per_step'], d['roofline'].get('frac_mfma', d['roofline'].get('frac')), d['roofline'].get('frac_hbm')))
except Exception as e:
    print('parse failed $B $M', e)
PY
      done; done ;;
    *) bash $R/tools/gpu_r5.sh $TAG $(left) $STEP ;;
  esac
done
echo "=== done in $(( $(date +%s) - T0 )) s"
