#!/bin/bash
# Round-5 GPU session runner (through gpurun).  New steps live here; every other step name is handed to tools/gpu_r4.sh.
# usage: tools/gpu_r5.sh <tag> <deadline_s> <step> [<step> ...]
#   steps: tests_r5 mlp_vars (VARS='name ...', FORMS='auto phased') seqmean dense ... + gpu_r4.sh's
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
line() {  # file label: one line of a bench JSON
python - <<PY
import json
try:
    d = json.loads(open('$1').read().strip().splitlines()[-1])
    r = d['roofline']
    print('%-28s qps %9.0f ms/step %7.4f kernel_ms %7.4f frac %s valid %s parity %s' % ('$2', d['value'], d['ms_per_step'], r['kernel_ms'], r.get('frac'), d.get('valid_queries'), d.get('parity')))
except Exception as e:
    print('$2 parse failed', e)
PY
}
for STEP in "$@"; do
  if [ $(left) -lt 45 ]; then echo "SKIP $STEP (deadline)"; continue; fi
  echo "=== $STEP (left $(left) s)"
  case $STEP in
    tests_r5)  # what round 5 added
      timeout 900 python -m pytest tests/test_hash_boundary_gpu.py tests/test_ops_gpu.py tests/test_search_gpu.py tests/test_zz_baseline_configs_gpu.py -m gpu -q --timeout 600 \
          -k "${TESTS_K:-last_id or user_seq_mean or group_gather or dense}" > $OUT/pytest_r5_$TAG.log 2>&1
      tail -4 $OUT/pytest_r5_$TAG.log; grep -E "^(E  |FAILED|ERROR)" $OUT/pytest_r5_$TAG.log | head -30 ;;
    mlp_vars)  # steady-state A/B of library variants on configs[2] (split-f16, batch 1024): fused kernel and pipeline of phases
      for L in $R/nann_amd/_build/libnann_hip.so $R/nann_amd/_build/var_*/libnann_hip.so; do
        V=$(basename $(dirname $L)); [ "$V" = "_build" ] && V=shipped
        if [ -n "${VARS:-}" ] && ! echo " $VARS shipped " | grep -q " ${V#var_} "; then continue; fi
        for M in ${FORMS:-auto phased}; do
          if [ $(left) -lt 60 ]; then echo "SKIP $V $M"; continue; fi
          NANN_MLP_FORM=$M NANN_HIP_LIB=$L timeout 200 $BENCH --scorer mlp --batch ${VAR_BATCH:-1024} --steps ${VAR_STEPS:-150} --warmup ${VAR_WARMUP:-100} --no-secondary --no-cpu-baseline > $OUT/mlpvar_${V}_m${M}_$TAG.json 2> $OUT/mlpvar_${V}_m${M}_$TAG.err
          line $OUT/mlpvar_${V}_m${M}_$TAG.json "VAR ${V#var_} form $M"
        done
      done ;;
    seqmean)  # the headline step with the new k_user_seq_mean: step - kernel, then the kernel's own duration under rocprofv3
      timeout 300 $BENCH --no-secondary --no-cpu-baseline --steps 20 > $OUT/bench_l2_$TAG.json 2> $OUT/bench_l2_$TAG.err
      line $OUT/bench_l2_$TAG.json "L2 headline"
      rm -rf /tmp/prof/kt_l2
      ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt_l2 -o kt -- $BENCH --steps 10 --warmup 2 --no-secondary --no-cpu-baseline > $OUT/prof_kt_l2_$TAG.log 2>&1 )
      find /tmp/prof/kt_l2 -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_l2_$TAG.csv \;
      grep -E "k_search|k_user_seq" $OUT/kernel_stats_l2_$TAG.csv | cut -c1-200 ;;
    dense)  # configs[1] on the dense graph family (keepPrunedConnections): plan, reruns, rate, roofline, parity
      timeout 400 $BENCH --graph hnsw_dense --no-secondary --no-cpu-baseline --steps 10 > $OUT/bench_dense_$TAG.json 2> $OUT/bench_dense_$TAG.err
      line $OUT/bench_dense_$TAG.json "L2 dense graph"
      python - <<PY
import json
d = json.loads(open('$OUT/bench_dense_$TAG.json').read().strip().splitlines()[-1])
print('   mean_degree_l0', d.get('mean_degree_l0'), 'plan', d.get('plan'), 'reruns', d.get('reruns_last_step'), 'recall', d.get('recall_at_k_vs_bruteforce'),
      'rows/q', d['roofline'].get('rows_scored_per_query'), 'gathered/q', d['roofline'].get('gathered_per_query'))
PY
      for M in lds_hash lds_hash32; do
        timeout 300 $BENCH --graph hnsw_dense --no-secondary --no-cpu-baseline --steps 10 --traversal $M > $OUT/bench_dense_${M}_$TAG.json 2> $OUT/bench_dense_${M}_$TAG.err
        line $OUT/bench_dense_${M}_$TAG.json "L2 dense graph $M"
        python -c "import json;d=json.loads(open('$OUT/bench_dense_${M}_$TAG.json').read().strip().splitlines()[-1]);print('   plan',d.get('plan'),'reruns',d.get('reruns_last_step'))"
      done ;;
    reserve_sweep)  # the slot reserve under an exchange as long as xGMI's (loopback copies x 15), one GPU
      timeout 600 python tools/overlap_bench.py /tmp/idx 30 --repeat 15 --reserves 0,8,16,32 > $OUT/reserve_sweep_$TAG.jsonl 2> $OUT/reserve_sweep_$TAG.err
      timeout 300 python tools/overlap_bench.py /tmp/idx 30 --repeat 1 --reserves 0,16 >> $OUT/reserve_sweep_$TAG.jsonl 2>> $OUT/reserve_sweep_$TAG.err
      python - <<PY
import json
for l in open('$OUT/reserve_sweep_$TAG.jsonl'):
    d = json.loads(l); print(d['summary'], d['exchange_parts_ms'])
PY
      tail -2 $OUT/reserve_sweep_$TAG.err | grep -v amdgpu.ids ;;
    *)
      bash $R/tools/gpu_r4.sh $TAG $(left) $STEP ;;
  esac
done
echo "done left=$(left)"
