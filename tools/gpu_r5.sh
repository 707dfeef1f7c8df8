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
    knn)  # configs[1] on the exact k-NN graph (every level-0 row at the cap of 64): plan, reruns, rate; then both hash plans forced
      for M in auto lds_hash lds_hash32; do
        timeout 300 $BENCH --graph knn --no-secondary --no-cpu-baseline --steps 10 --traversal $M > $OUT/bench_knn_${M}_$TAG.json 2> $OUT/bench_knn_${M}_$TAG.err
        line $OUT/bench_knn_${M}_$TAG.json "L2 knn graph $M"
        python -c "import json;d=json.loads(open('$OUT/bench_knn_${M}_$TAG.json').read().strip().splitlines()[-1]);print('   plan',d.get('plan'),'reruns',d.get('reruns_last_step'),'probe',d.get('index_probe'),'recall',d.get('recall_at_k_vs_bruteforce'),'rows/q',d['roofline'].get('rows_scored_per_query'))"
      done ;;
    reserve_sweep)  # the slot reserve under an exchange as long as xGMI's (loopback copies x 15), one GPU
      timeout 600 python tools/overlap_bench.py /tmp/idx 30 --wait-us ${SWEEP_WAIT_US:-1400} --reserves 0,8,16,32,64 > $OUT/reserve_sweep_$TAG.jsonl 2> $OUT/reserve_sweep_$TAG.err
      timeout 300 python tools/overlap_bench.py /tmp/idx 30 --repeat 1 --reserves 0,16 >> $OUT/reserve_sweep_$TAG.jsonl 2>> $OUT/reserve_sweep_$TAG.err
      python - <<PY
import json
for l in open('$OUT/reserve_sweep_$TAG.jsonl'):
    d = json.loads(l); print(d['summary'], d['exchange_parts_ms'])
PY
      tail -2 $OUT/reserve_sweep_$TAG.err | grep -v amdgpu.ids ;;
    prof_r5)  # rocprofv3 evidence of the round-5 bench lines: kernel stats per workload, then separate --pmc passes (one counter set each)
      NOTE="round 5 ($TAG)"
      kst() {  # name -- command: rocprofv3 --kernel-trace --stats
        local NAME=$1; shift 2
        rm -rf /tmp/prof/kt_$NAME
        ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt_$NAME -o kt -- "$@" > $OUT/prof_kt_${NAME}_$TAG.log 2>&1 )
        find /tmp/prof/kt_$NAME -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_${NAME}_$TAG.csv \;
        grep -E "k_search|k_mlp_phase|k_user_seq" $OUT/kernel_stats_${NAME}_$TAG.csv | head -6 | cut -c1-170
      }
      pm() {  # name "tag tag" calls "workload" counters... -- command
        local NAME=$1 TAGS=$2 CALLS=$3 WL=$4; shift 4
        local CTRS=""
        while [ "$1" != "--" ]; do CTRS="$CTRS $1"; shift; done
        shift
        rm -rf /tmp/prof/pmc_$NAME
        ( cd /tmp && timeout 400 rocprofv3 --pmc $CTRS --output-format csv -d /tmp/prof/pmc_$NAME -o pmc -- "$@" > $OUT/prof_pmc_${NAME}_$TAG.log 2>&1 )
        local TA=""; for T in $TAGS; do TA="$TA --tag $T"; done
        python tools/pmc_r5.py /tmp/prof/pmc_$NAME $TA --calls $CALLS --workload "$WL" --note "$NOTE" | tee -a $OUT/pmc_r5_$TAG.txt
      }
      Q="--no-secondary --no-cpu-baseline --steps 3 --warmup 1"
      L2T="1000000x128f16_ef128_k200_b4096_l2_hnsw"
      for W in ${PROF_WORKLOADS:-l2 dense stress mlp mlp4m}; do
        if [ $(left) -lt 200 ]; then echo "SKIP prof $W"; continue; fi
        case $W in
          l2) kst l2 -- $BENCH --no-secondary --no-cpu-baseline --steps 10 --warmup 2
              pm l2_f "$L2T" 4 "BASELINE configs[1]: 1M x 128-d f16, ef=128, top-200, L2, batch 4096" FETCH_SIZE -- $BENCH $Q
              pm l2_w "$L2T" 4 "BASELINE configs[1]: 1M x 128-d f16, ef=128, top-200, L2, batch 4096" WRITE_SIZE -- $BENCH $Q
              pm l2_a "$L2T" 4 "BASELINE configs[1]: 1M x 128-d f16, ef=128, top-200, L2, batch 4096" SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -- $BENCH $Q
              pm l2_b "$L2T" 4 "BASELINE configs[1]: 1M x 128-d f16, ef=128, top-200, L2, batch 4096" GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $BENCH $Q ;;
          dense) kst dense -- $BENCH --graph hnsw_dense --no-secondary --no-cpu-baseline --steps 10 --warmup 2
              pm dense_f "${L2T}_dense" 4 "configs[1] on the dense graph family (keepPrunedConnections)" FETCH_SIZE -- $BENCH --graph hnsw_dense $Q
              pm dense_w "${L2T}_dense" 4 "configs[1] on the dense graph family (keepPrunedConnections)" WRITE_SIZE -- $BENCH --graph hnsw_dense $Q ;;
          stress) S="--items 4000000 --dim 256 --dtype bf16 --ef 256 --batch 2048"
              kst stress4m -- $BENCH $S --no-secondary --no-cpu-baseline --steps 5 --warmup 2
              pm stress_f "4000000x256bf16_ef256 4000000x256bf16_ef256_k200_b2048_l2_hnsw" 4 "configs[4]'s shard: 4M x 256-d bf16, ef=256, L2, batch 2048" FETCH_SIZE -- $BENCH $S $Q
              pm stress_w "4000000x256bf16_ef256 4000000x256bf16_ef256_k200_b2048_l2_hnsw" 4 "configs[4]'s shard: 4M x 256-d bf16, ef=256, L2, batch 2048" WRITE_SIZE -- $BENCH $S $Q ;;
          mlp) M="--scorer mlp --batch 1024"
              MT="1000000x128f16_ef128_k200_b1024_mlp_hnsw ${L2T}_mlp_split"
              kst mlp -- $BENCH $M --no-secondary --no-cpu-baseline --steps 20 --warmup 40
              pm mlp_a "$MT" 4 "BASELINE configs[2]: MLP 256-128-1 split-f16, batch 1024" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -- $BENCH $M $Q
              pm mlp_b "$MT" 4 "BASELINE configs[2]: MLP 256-128-1 split-f16, batch 1024" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS -- $BENCH $M $Q
              pm mlp_f "$MT" 4 "BASELINE configs[2]: MLP 256-128-1 split-f16, batch 1024" FETCH_SIZE -- $BENCH $M $Q
              pm mlp_w "$MT" 4 "BASELINE configs[2]: MLP 256-128-1 split-f16, batch 1024" WRITE_SIZE -- $BENCH $M $Q ;;
          mlp4m) M="--items 4000000 --dim 256 --dtype bf16 --ef 256 --scorer mlp --batch 1024"
              MT="4000000x256bf16_ef256_mlp_split 4000000x256bf16_ef256_k200_b1024_mlp_hnsw"
              kst mlp4m -- $BENCH $M --no-secondary --no-cpu-baseline --steps 10 --warmup 20
              pm mlp4m_a "$MT" 4 "configs[4]'s shard under the MLP (split-f16, pipeline of phases on the 32K-slot plan), batch 1024" SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- $BENCH $M $Q
              pm mlp4m_b "$MT" 4 "configs[4]'s shard under the MLP (split-f16, pipeline of phases on the 32K-slot plan), batch 1024" GRBM_GUI_ACTIVE -- $BENCH $M $Q
              pm mlp4m_f "$MT" 4 "configs[4]'s shard under the MLP (split-f16, pipeline of phases on the 32K-slot plan), batch 1024" FETCH_SIZE -- $BENCH $M $Q
              pm mlp4m_w "$MT" 4 "configs[4]'s shard under the MLP (split-f16, pipeline of phases on the 32K-slot plan), batch 1024" WRITE_SIZE -- $BENCH $M $Q ;;
        esac
      done
      cp $R/profiles/pmc_latest.json $OUT/pmc_latest_$TAG.json ;;
    *)
      bash $R/tools/gpu_r4.sh $TAG $(left) $STEP ;;
  esac
done
echo "done left=$(left)"
