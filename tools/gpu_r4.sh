#!/bin/bash
# Round-4 GPU session runner (through gpurun): named steps, each skipped once the deadline has passed.
# (Round 5 retired NANN_MLP_MAPPING -- the steps below that set it now compare a form with itself; NANN_MLP_FORM=fused|phased
#  is what is left of it, tools/gpu_r5.sh mlp_vars.  The generic steps -- tests_full, smoke, bench_full, prof_default, dry8,
#  serve, eval_bench ... -- are what tools/gpu_r5.sh still hands over to this file.)
# usage: tools/gpu_r4.sh <tag> <deadline_s> <step> [<step> ...]
#   steps: tests_full tests_mlp tests_new tests_k (TESTS_K='<-k expr>') smoke bench_full bench_l2 bench_mlp bench_mlp_ab bench_mlp_maps
#          bench_mlp_wide mlp_exact_diag bench_attn b1 b1modes phase_4m rate_mlp serve prof_l2 prof_mlp prof_attn prof_stress prof_4m
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
show() {  # file label
python - <<PY
import json
try:
    d = json.loads(open('$1').read().strip().splitlines()[-1])
    r = d['roofline']
    print('$2', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'issued', r.get('issued_frac_of_peak'),
          'valid', d.get('valid_queries'), 'cpu', d.get('cpu_baseline', {}).get('value'), 'parity', d.get('parity'),
          'recall', d.get('recall_at_k_vs_bruteforce'), 'setup_s', d.get('setup_s'), 'rows/q', r.get('rows_scored_per_query'))
    if 'phase_breakdown' in d:
        print('   ticks', {k: round(v) for k, v in d['phase_breakdown']['ticks_per_query'].items() if v})
    for k, v in d.get('secondary', {}).items():
        if isinstance(v, dict) and 'roofline' in v:
            print('   SEC', k, 'qps', v['qps_end_to_end'], 'kernel_ms', v['roofline']['kernel_ms'], 'frac', v['roofline']['frac'],
                  'valid', v.get('valid_queries'), 'parity', v.get('parity'), 'recall', v.get('recall_at_k_vs_bruteforce'),
                  'cpu', v.get('cpu_baseline', {}).get('value'), 'setup_s', v.get('setup_s'))
        else:
            print('   SEC', k, v)
except Exception as e:
    print('$2 parse failed', e)
PY
}
pmc() {  # name, kernel substring, counters..., then -- command
  local NAME=$1 KSUB=$2; shift 2
  local CTRS=""
  while [ "$1" != "--" ]; do CTRS="$CTRS $1"; shift; done
  shift
  rm -rf /tmp/prof/pmc_$NAME
  ( cd /tmp && timeout 240 rocprofv3 --pmc $CTRS --output-format csv -d /tmp/prof/pmc_$NAME -o pmc -- "$@" > $OUT/prof_pmc_${NAME}_$TAG.log 2>&1 )
  python - <<PY >> $OUT/pmc_$TAG.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/prof/pmc_$NAME/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if '$KSUB' in r.get('Kernel_Name', ''):
            acc[(r['Kernel_Name'][:64], r.get('Counter_Name'))].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    print('$NAME', c, k, 'dispatches', len(v), 'mean', sum(v) / len(v), 'min', min(v), 'max', max(v))
PY
}
kstats() {  # name -- command
  local NAME=$1; shift 2
  rm -rf /tmp/prof/kt_$NAME
  ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt_$NAME -o kt -- "$@" > $OUT/prof_kt_${NAME}_$TAG.log 2>&1 )
  find /tmp/prof/kt_$NAME -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_${NAME}_$TAG.csv \;
  grep -E "k_search|k_score|k_hnsw" $OUT/kernel_stats_${NAME}_$TAG.csv | head -4
}
: > $OUT/pmc_$TAG.txt
for STEP in "$@"; do
  if [ $(left) -lt 45 ]; then echo "SKIP $STEP (deadline)"; continue; fi
  echo "=== $STEP (left $(left) s)"
  case $STEP in
    tests_mlp)
      timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_search_gpu.py -m gpu -q --timeout 300 -x \
          -k "mlp or frozen or model_directory or serving_signature or batch_size" > $OUT/pytest_mlp_$TAG.log 2>&1
      tail -4 $OUT/pytest_mlp_$TAG.log; grep -E "^(E  |FAILED|ERROR)" $OUT/pytest_mlp_$TAG.log | head -20 ;;
    tests_full)
      timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_$TAG.log 2>&1
      tail -4 $OUT/pytest_$TAG.log; grep -E "^(E  |FAILED|ERROR)" $OUT/pytest_$TAG.log | head -30 ;;
    tests_new)
      timeout 900 python -m pytest tests/test_index_build_gpu.py tests/test_zz_baseline_configs_gpu.py tests/test_search_gpu.py -m gpu -q --timeout 600 \
          -k "device_builder or metric or config0 or config3_real or cpp_serving or serving_front" > $OUT/pytest_new_$TAG.log 2>&1
      tail -4 $OUT/pytest_new_$TAG.log; grep -E "^(E  |FAILED|ERROR)" $OUT/pytest_new_$TAG.log | head -20 ;;
    serve)
      timeout 400 python tools/serve_bench.py 1000000 64 512 2048 2> $OUT/serve_$TAG.err | tee $OUT/serve_$TAG.txt; tail -2 $OUT/serve_$TAG.err ;;
    b1)
      for B in 1 64; do
        timeout 200 $BENCH --batch $B --steps 20 --warmup 3 --no-secondary --no-cpu-baseline --phase-ticks > $OUT/bench_b${B}_$TAG.json 2> $OUT/bench_b${B}_$TAG.err
        show $OUT/bench_b${B}_$TAG.json "B=$B"
      done ;;
    b1modes)
      for M in lds_hash32 lds_bitmap; do for B in 1 64; do
        timeout 200 $BENCH --batch $B --steps 20 --warmup 3 --no-secondary --no-cpu-baseline --phase-ticks --traversal $M > $OUT/bench_b${B}_${M}_$TAG.json 2> $OUT/bench_b${B}_${M}_$TAG.err
        show $OUT/bench_b${B}_${M}_$TAG.json "B=$B $M"
      done; done ;;
    smoke)
      timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; tail -1 $OUT/smoke_$TAG.log ;;
    rate_mlp)
      timeout 200 python tools/mlp_rate.py 128 3145728 split > $OUT/rate_mlp_$TAG.txt 2>&1
      NANN_MLP_MAPPING=1 timeout 200 python tools/mlp_rate.py 128 3145728 split >> $OUT/rate_mlp_$TAG.txt 2>&1
      timeout 200 python tools/mlp_rate.py 64 3145728 split >> $OUT/rate_mlp_$TAG.txt 2>&1
      cat $OUT/rate_mlp_$TAG.txt ;;
    bench_mlp)
      timeout 400 $BENCH --scorer mlp --batch 1024 --steps 5 --warmup 2 --no-secondary --phase-ticks --cpu-seconds 6 > $OUT/bench_mlp_$TAG.json 2> $OUT/bench_mlp_$TAG.err
      show $OUT/bench_mlp_$TAG.json MLP; tail -2 $OUT/bench_mlp_$TAG.err ;;
    bench_mlp_maps)
      for M in 2 4; do
        NANN_MLP_MAPPING=$M timeout 300 $BENCH --scorer mlp --batch 1024 --steps 5 --warmup 2 --no-secondary --phase-ticks --no-cpu-baseline > $OUT/bench_mlp_map${M}_$TAG.json 2> $OUT/bench_mlp_map${M}_$TAG.err
        show $OUT/bench_mlp_map${M}_$TAG.json "MLP_MAPPING_$M"
      done ;;
    bench_mlp_ab)
      NANN_MLP_MAPPING=1 timeout 300 $BENCH --scorer mlp --batch 1024 --steps 5 --warmup 2 --no-secondary --phase-ticks --no-cpu-baseline > $OUT/bench_mlp_map1_$TAG.json 2> $OUT/bench_mlp_map1_$TAG.err
      show $OUT/bench_mlp_map1_$TAG.json MLP_FIRST_MAPPING ;;
    bench_l2)
      timeout 300 $BENCH --no-secondary --phase-ticks --no-cpu-baseline --steps 10 > $OUT/bench_l2_$TAG.json 2> $OUT/bench_l2_$TAG.err
      show $OUT/bench_l2_$TAG.json L2; tail -2 $OUT/bench_l2_$TAG.err ;;
    bench_full)
      timeout 900 $BENCH --phase-ticks --cpu-seconds 8 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
      show $OUT/bench_$TAG.json DEFAULT; tail -3 $OUT/bench_$TAG.err ;;
    prof_mlp)
      kstats mlp -- $BENCH --scorer mlp --batch 1024 --steps 5 --warmup 1 --no-secondary --no-cpu-baseline
      pmc mlp_a k_search SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -- $BENCH --scorer mlp --batch 1024 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline
      pmc mlp_b k_search GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $BENCH --scorer mlp --batch 1024 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline ;;
    prof_l2)
      kstats l2 -- $BENCH --steps 10 --warmup 2 --no-secondary --no-cpu-baseline
      pmc l2_fetch k_search FETCH_SIZE -- $BENCH --steps 3 --warmup 1 --no-secondary --no-cpu-baseline
      pmc l2_write k_search WRITE_SIZE -- $BENCH --steps 3 --warmup 1 --no-secondary --no-cpu-baseline ;;
    prof_stress)
      S="--items 2000000 --dim 256 --dtype bf16 --ef 256 --batch 2048 --no-secondary --no-cpu-baseline"
      kstats stress -- $BENCH $S --steps 5 --warmup 2
      pmc stress_fetch k_search FETCH_SIZE -- $BENCH $S --steps 3 --warmup 1
      pmc stress_write k_search WRITE_SIZE -- $BENCH $S --steps 3 --warmup 1 ;;
    prof_4m)
      S="--items 4000000 --dim 256 --dtype bf16 --ef 256 --batch 2048 --no-secondary"
      timeout 400 $BENCH $S --steps 5 --warmup 2 --cpu-seconds 6 > $OUT/bench_4m_$TAG.json 2> $OUT/bench_4m_$TAG.err
      show $OUT/bench_4m_$TAG.json SHARD_4M
      kstats shard4m -- $BENCH $S --steps 5 --warmup 2 --no-cpu-baseline
      pmc shard4m_fetch k_search FETCH_SIZE -- $BENCH $S --steps 3 --warmup 1 --no-cpu-baseline
      pmc shard4m_write k_search WRITE_SIZE -- $BENCH $S --steps 3 --warmup 1 --no-cpu-baseline ;;
    tests_k)  # TESTS_K='<-k expression>' tools/gpu_r4.sh ...
      timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -k "${TESTS_K:-mlp}" > $OUT/pytest_k_$TAG.log 2>&1
      tail -4 $OUT/pytest_k_$TAG.log; grep -E "^(E  |FAILED|ERROR)" $OUT/pytest_k_$TAG.log | head -30 ;;
    bench_mlp_wide)  # split-f16 MLP on config 5's row shape and beam: 1.2M x 256-d bf16, ef=256 (bitmap plan + pre-projected scorer)
      S="--items 1200000 --dim 256 --dtype bf16 --ef 256 --scorer mlp --batch 1024 --steps 4 --warmup 1 --no-secondary --no-cpu-baseline"
      timeout 400 $BENCH $S > $OUT/bench_mlp_wide_$TAG.json 2> $OUT/bench_mlp_wide_$TAG.err
      show $OUT/bench_mlp_wide_$TAG.json MLP_WIDE; tail -2 $OUT/bench_mlp_wide_$TAG.err
      NANN_MLP_MAPPING=1 timeout 400 $BENCH $S > $OUT/bench_mlp_wide_map1_$TAG.json 2> $OUT/bench_mlp_wide_map1_$TAG.err
      show $OUT/bench_mlp_wide_map1_$TAG.json MLP_WIDE_FIRST_MAPPING ;;
    mlp_exact_diag)
      timeout 300 $BENCH --scorer mlp --mlp-precision exact --batch 1024 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > $OUT/bench_mlp_exact_$TAG.json 2> $OUT/bench_mlp_exact_$TAG.err
      show $OUT/bench_mlp_exact_$TAG.json MLP_EXACT
      python -c "import json;d=json.loads(open('$OUT/bench_mlp_exact_$TAG.json').read().strip().splitlines()[-1]);print('host_enqueue_ms',d.get('host_enqueue_ms'))" ;;
    bench_attn)
      timeout 300 python tools/attn_bench.py /tmp/idx 512 1024 > $OUT/bench_attn_$TAG.txt 2> $OUT/bench_attn_$TAG.err
      NANN_MLP_MAPPING=1 timeout 300 python tools/attn_bench.py /tmp/idx 512 >> $OUT/bench_attn_$TAG.txt 2>> $OUT/bench_attn_$TAG.err
      cat $OUT/bench_attn_$TAG.txt; tail -2 $OUT/bench_attn_$TAG.err ;;
    prof_attn)
      kstats attn -- python $R/tools/attn_bench.py /tmp/idx 512
      pmc attn_a k_search SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -- python $R/tools/attn_bench.py /tmp/idx 512
      pmc attn_b k_search GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- python $R/tools/attn_bench.py /tmp/idx 512 ;;
    phase_4m)
      S="--items 4000000 --dim 256 --dtype bf16 --ef 256 --batch 2048 --no-secondary --no-cpu-baseline --phase-ticks"
      timeout 400 $BENCH $S --steps 3 --warmup 1 > $OUT/bench_4m_phase_$TAG.json 2> $OUT/bench_4m_phase_$TAG.err
      show $OUT/bench_4m_phase_$TAG.json SHARD_4M_PHASES ;;
    prof_small)  # one launch of 1 and of 64 queries: kernel time by rocprofv3 next to the HIP-event numbers of the batch sweep
      for B in 1 64; do
        kstats b$B -- $BENCH --batch $B --steps 30 --warmup 5 --no-secondary --no-cpu-baseline
      done ;;
    prof_mlp_wide)
      S="--items 2000000 --dim 256 --dtype bf16 --ef 256 --scorer mlp --batch 1024 --steps 4 --warmup 1 --no-secondary --no-cpu-baseline"
      kstats mlpwide -- $BENCH $S
      pmc mlpwide_a k_search SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU -- $BENCH $S
      pmc mlpwide_b k_search GRBM_GUI_ACTIVE -- $BENCH $S ;;
    tests_r4)  # what round 4 changed: resident-layer-2 MLP scorers (both precisions), per-query level_topn, table lifecycle
      timeout 1200 python -m pytest tests/test_search_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout 600 -x \
          -k "mlp or level_topn or prepared or concurrent or preprojection or projection_tables or eval_graph or serving_signature or model_directory or frozen or batch_size or committed" > $OUT/pytest_r4_$TAG.log 2>&1
      tail -4 $OUT/pytest_r4_$TAG.log; grep -E "^(E  |FAILED|ERROR)" $OUT/pytest_r4_$TAG.log | head -30 ;;
    tests_cfg)  # BASELINE-config shapes under the MLP (1M, 1.2M wide beam, 4M)
      timeout 1500 python -m pytest tests/test_zz_baseline_configs_gpu.py -m gpu -q --timeout 900 -x -k "mlp" > $OUT/pytest_cfg_$TAG.log 2>&1
      tail -4 $OUT/pytest_cfg_$TAG.log; grep -E "^(E  |FAILED|ERROR)" $OUT/pytest_cfg_$TAG.log | head -30 ;;
    bench_mlp_ab5)  # same box: layer 2 resident in LDS (default, mapping 5) vs round 3's streamed slices (mapping 3)
      for M in 5 3; do
        NANN_MLP_MAPPING=$M timeout 300 $BENCH --scorer mlp --batch 1024 --steps 5 --warmup 2 --no-secondary --phase-ticks --no-cpu-baseline > $OUT/bench_mlp_map${M}_$TAG.json 2> $OUT/bench_mlp_map${M}_$TAG.err
        show $OUT/bench_mlp_map${M}_$TAG.json "MLP_SPLIT_MAPPING_$M"; tail -2 $OUT/bench_mlp_map${M}_$TAG.err
      done ;;
    bench_mlp_exact_ab)  # exact f32: pre-projected + resident (mapping 5) vs all layers on the f32 MFMA (mapping 3)
      for M in 5 3; do
        NANN_MLP_MAPPING=$M timeout 300 $BENCH --scorer mlp --mlp-precision exact --batch 1024 --steps 3 --warmup 1 --no-secondary --phase-ticks --no-cpu-baseline > $OUT/bench_mlp_exact_map${M}_$TAG.json 2> $OUT/bench_mlp_exact_map${M}_$TAG.err
        show $OUT/bench_mlp_exact_map${M}_$TAG.json "MLP_EXACT_MAPPING_$M"; tail -2 $OUT/bench_mlp_exact_map${M}_$TAG.err
      done ;;
    prof_mlp4)  # kernel stats + counters of the split-f16 MLP traversal, incl. the HBM traffic passes
      S="--scorer mlp --batch 1024 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline"
      kstats mlp -- $BENCH --scorer mlp --batch 1024 --steps 5 --warmup 1 --no-secondary --no-cpu-baseline
      pmc mlp_a k_search SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -- $BENCH $S
      pmc mlp_b k_search GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $BENCH $S
      pmc mlp_fetch k_search FETCH_SIZE -- $BENCH $S
      pmc mlp_write k_search WRITE_SIZE -- $BENCH $S ;;
    prof_mlp_exact)
      S="--scorer mlp --mlp-precision exact --batch 1024 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline"
      kstats mlpx -- $BENCH $S
      pmc mlpx_a k_search SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU -- $BENCH $S
      pmc mlpx_b k_search GRBM_GUI_ACTIVE -- $BENCH $S
      pmc mlpx_fetch k_search FETCH_SIZE -- $BENCH $S
      pmc mlpx_write k_search WRITE_SIZE -- $BENCH $S ;;
    prof_attn4)
      kstats attn -- python $R/tools/attn_bench.py /tmp/idx 512
      pmc attn_a k_search SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -- python $R/tools/attn_bench.py /tmp/idx 512
      pmc attn_b k_search GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS -- python $R/tools/attn_bench.py /tmp/idx 512
      pmc attn_fetch k_search FETCH_SIZE -- python $R/tools/attn_bench.py /tmp/idx 512
      pmc attn_write k_search WRITE_SIZE -- python $R/tools/attn_bench.py /tmp/idx 512 ;;
    bench_vars)  # every tuning / timing build under nann_amd/_build/var_*/ (tools/build_res_variant.py) + the shipped one, same box
      for L in $R/nann_amd/_build/libnann_hip.so $R/nann_amd/_build/var_*/libnann_hip.so; do
        V=$(basename $(dirname $L)); [ "$V" = "_build" ] && V=shipped
        for P in ${VAR_PRECISIONS:-split exact}; do
          NANN_HIP_LIB=$L timeout 200 $BENCH --scorer mlp --mlp-precision $P --batch 1024 --steps ${VAR_STEPS:-3} --warmup ${VAR_WARMUP:-1} --no-secondary --phase-ticks --no-cpu-baseline > $OUT/bench_var_${V}_${P}_$TAG.json 2> $OUT/bench_var_${V}_${P}_$TAG.err
          python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_var_${V}_${P}_$TAG.json').read().strip().splitlines()[-1])
    t = d['phase_breakdown']['ticks_per_query']
    rows = d['roofline']['rows_scored_per_query']
    tot = sum(t[k] for k in ('zero', 'walk', 'expand', 'score', 'topk', 'other'))
    print('VAR %-14s %-5s qps %9.0f kernel_ms %7.3f rows/q %6.0f score_ticks/row %6.1f (MFMA floor %s) other_ticks/q %7.0f clk %.2f GHz parity %s' % ('$V', '$P', d['value'], d['roofline']['kernel_ms'], rows, t['score'] / rows, '48' if '$P' == 'split' else '256', tot - t['score'], tot * 4 / d['roofline']['kernel_ms'] / 1e6, d.get('parity', {}).get('ids_identical', d.get('parity', {}).get('ids_equal'))))
except Exception as e:
    print('VAR $V $P failed', e)
PY
        done
      done ;;
    power)  # board power + shader clock sampled (rocm-smi) while a workload runs long enough to be sampled: is the MLP traversal at the power limit?
      rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -v "^=" | head -30 > $OUT/power_idle_$TAG.txt
      for W in "l2 --batch 4096 --steps 2500" "mlp_split --scorer mlp --batch 1024 --steps 2500" "mlp_exact --scorer mlp --mlp-precision exact --batch 1024 --steps 1000"; do
        set -- $W; N=$1; shift
        ( timeout 200 $BENCH "$@" --warmup 3 --no-secondary --no-cpu-baseline > $OUT/bench_power_${N}_$TAG.json 2> $OUT/bench_power_${N}_$TAG.err ) &
        BP=$!
        : > $OUT/power_${N}_$TAG.txt
        while kill -0 $BP 2>/dev/null; do
          rocm-smi --showpower --showclocks --json 2>/dev/null >> $OUT/power_${N}_$TAG.txt; echo >> $OUT/power_${N}_$TAG.txt
          sleep 0.15
        done
        wait $BP
        python - <<PY
import json, re
pw, ck = [], []
for line in open('$OUT/power_${N}_$TAG.txt'):
    line = line.strip()
    if not line.startswith('{'): continue
    try: d = json.loads(line)
    except ValueError: continue
    for card, v in d.items():
        for k, x in v.items():
            if 'Power' in k and 'W' in k:
                try: pw.append(float(x))
                except ValueError: pass
            if k.startswith('sclk'):
                m = re.search(r'(\d+)Mhz', str(x))
                if m: ck.append(int(m.group(1)))
try:
    d = json.loads(open('$OUT/bench_power_${N}_$TAG.json').read().strip().splitlines()[-1])
    extra = 'qps %.0f kernel_ms %.3f' % (d['value'], d['roofline']['kernel_ms'])
except Exception as e:
    extra = 'bench failed %s' % e
pw.sort(); ck.sort()
top = pw[len(pw) // 2:] if pw else [0]
print('POWER %-10s samples %3d power W: median-of-upper-half %.0f max %.0f | sclk MHz upper-half-median %s min %s | %s' % (
    '$N', len(pw), top[len(top) // 2], max(pw or [0]), ck[len(ck) * 3 // 4] if ck else None, min(ck) if ck else None, extra))
PY
      done
      head -12 $OUT/power_idle_$TAG.txt ;;
    overlap)  # item 7a: does the exchange overlap the next batch's search on one GPU (8-shard loopback), with and without reserved slots
      for RSV in 0 8 32; do
        NANN_SEARCH_SLOT_RESERVE=$RSV timeout 300 python tools/overlap_bench.py /tmp/idx 30 > $OUT/overlap_rsv${RSV}_$TAG.json 2> $OUT/overlap_rsv${RSV}_$TAG.err
        python -c "import json;d=json.loads(open('$OUT/overlap_rsv${RSV}_$TAG.json').read().strip().splitlines()[-1]);print('OVERLAP reserve $RSV', d['summary'])" || tail -3 $OUT/overlap_rsv${RSV}_$TAG.err
      done
      rm -rf /tmp/prof/kt_overlap
      ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof/kt_overlap -o kt -- python $R/tools/overlap_bench.py /tmp/idx 8 > $OUT/overlap_trace_$TAG.log 2>&1 )
      python - <<PY
import csv, glob
rows = []
for f in glob.glob('/tmp/prof/kt_overlap/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]))
rows.sort()
srch = [(a, b) for a, b, n in rows if 'k_search' in n]
mrg = [(a, b, n) for a, b, n in rows if 'k_merge_records' in n or 'k_pack_record' in n]
inside = sum(1 for a, b, n in mrg if any(sa < a and b < sb for sa, sb in srch))
partly = sum(1 for a, b, n in mrg if any(a < sb and sa < b for sa, sb in srch))
print('OVERLAP trace: exchange kernels', len(mrg), 'fully inside a k_search interval', inside, 'overlapping one', partly)
open('$OUT/overlap_trace_$TAG.txt', 'w').write('\n'.join('%d %d %s' % r for r in rows[-400:]))
PY
      ;;
    dry8)  # item 7b: the whole 8-rank flow of bench.py on ONE GPU (gloo, exchange staged through the host), merged parity vs the oracle
      NANN_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus 8 --dist-backend gloo --items 200000 --batch 1024 --steps 3 --warmup 1 --index-cache /tmp/idx --no-secondary --no-cpu-baseline > $OUT/bench_dry8_$TAG.json 2> $OUT/bench_dry8_$TAG.err
      tail -1 $OUT/bench_dry8_$TAG.json | cut -c1-900; tail -3 $OUT/bench_dry8_$TAG.err ;;
    bench_mlp_batches)  # configs[2] at batch 1024 / 2048 / 4096, steady state: the tail of a 4-queries-per-workgroup launch
      for B in 1024 2048 4096; do
        timeout 300 $BENCH --scorer mlp --batch $B --steps 60 --warmup 100 --no-secondary --phase-ticks --no-cpu-baseline > $OUT/bench_mlp_b${B}_$TAG.json 2> $OUT/bench_mlp_b${B}_$TAG.err
        show $OUT/bench_mlp_b${B}_$TAG.json "MLP_SPLIT_B$B"
      done ;;
    bench_mlp_phased)  # same box, steady state: the pipeline of phases (mapping 6, default) against the fused resident kernel (5), both precisions
      for P in split exact; do for M in 6 5; do
        NANN_MLP_MAPPING=$M timeout 300 $BENCH --scorer mlp --mlp-precision $P --batch ${MLP_BATCH:-1024} --steps 60 --warmup 100 --no-secondary --no-cpu-baseline > $OUT/bench_mlp_${P}_map${M}_$TAG.json 2> $OUT/bench_mlp_${P}_map${M}_$TAG.err
        show $OUT/bench_mlp_${P}_map${M}_$TAG.json "MLP_${P}_MAPPING_$M"; tail -2 $OUT/bench_mlp_${P}_map${M}_$TAG.err | grep -v amdgpu.ids
      done; done ;;
    prof_phased)  # per-launch durations of the pipeline of phases at steady-state clock, both precisions
      for P in split exact; do
        rm -rf /tmp/prof/kt_ph_$P
        ( cd /tmp && timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt_ph_$P -o kt -- $BENCH --scorer mlp --mlp-precision $P --batch ${MLP_BATCH:-1024} --steps 30 --warmup 60 --no-secondary --no-cpu-baseline > $OUT/prof_kt_ph_${P}_$TAG.log 2>&1 )
        find /tmp/prof/kt_ph_$P -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_phased_${P}_$TAG.csv \;
        F=$(find /tmp/prof/kt_ph_$P -name '*kernel_trace.csv' | head -1)
        echo "PHASED $P"; python tools/phase_trace.py $F 20 | tee $OUT/phase_trace_${P}_$TAG.txt
      done ;;
    phase_vars)  # timing builds of the split-f16 scoring launch (nann_mlp6.h NANN_PHASE_VAR) as dry launches behind the real ones
      for L in $R/nann_amd/_build/libnann_hip.so $R/nann_amd/_build/var_ph*/libnann_hip.so; do
        V=$(basename $(dirname $L)); [ "$V" = "_build" ] && V=shipped
        rm -rf /tmp/prof/kt_pv
        ( cd /tmp && NANN_PHASE_SHADOW=1 NANN_HIP_LIB=$L timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof/kt_pv -o kt -- $BENCH --scorer mlp --mlp-precision split --batch 1024 --steps 20 --warmup 40 --no-secondary --no-cpu-baseline > $OUT/prof_kt_pv_${V}_$TAG.log 2>&1 )
        F=$(find /tmp/prof/kt_pv -name '*kernel_trace.csv' | head -1)
        echo "PHASEVAR $V"; python tools/phase_trace.py $F 12 | tee $OUT/phase_vars_${V}_$TAG.txt | grep -E "phase_score|sum of"
      done ;;
    pmc_phased)  # counters of the pipeline of phases, per kernel (sums over all dispatches of the run; clock = GRBM cycles / duration)
      for P in ${PH_PRECISIONS:-split}; do
        S="--scorer mlp --mlp-precision $P --batch 1024 --steps 8 --warmup 24 --no-secondary --no-cpu-baseline"
        for PASS in "a SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
                    "b GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
                    "fetch FETCH_SIZE" "write WRITE_SIZE"; do
          set -- $PASS; N=$1; shift
          rm -rf /tmp/prof/pmcph
          ( cd /tmp && timeout 240 rocprofv3 --pmc "$@" --output-format csv -d /tmp/prof/pmcph -o pmc -- $BENCH $S > $OUT/prof_pmcph_${P}_${N}_$TAG.log 2>&1 )
          python - <<PY | tee -a $OUT/pmc_phased_${P}_$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(float); cnt = collections.Counter(); dur = collections.defaultdict(float)
for f in glob.glob('/tmp/prof/pmcph/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        k = 'k_mlp_phase_score' if 'k_mlp_phase_score' in k else 'k_mlp_phase_prefix' if 'phase_prefix' in k else ('k_search' + ('<phase>' if ', 9,' in k else '')) if 'k_search' in k else None
        if k is None: continue
        acc[(k, r['Counter_Name'])] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        if 'Start_Timestamp' in r and r['Counter_Name'] in ('GRBM_GUI_ACTIVE', 'SQ_BUSY_CYCLES', 'FETCH_SIZE', 'WRITE_SIZE'):
            dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
for (k, c), v in sorted(acc.items()):
    print('PMCPH $P $N', k, c, 'dispatches', cnt[(k, c)], 'sum', v, ('dur_ns %.0f' % dur[k]) if dur.get(k) else '')
PY
        done
      done ;;
    eval_bench)  # the evaluation traversal (f3) on the 1M index: users/s, both bitmap placements, oracle parity on a sample
      timeout 400 python tools/eval_bench.py /tmp/idx 1024 6 > $OUT/eval_bench_$TAG.json 2> $OUT/eval_bench_$TAG.err; tail -1 $OUT/eval_bench_$TAG.json | cut -c1-900; tail -2 $OUT/eval_bench_$TAG.err | grep -v amdgpu.ids
      NANN_EVAL_SEEN=hbm timeout 400 python tools/eval_bench.py /tmp/idx 1024 3 > $OUT/eval_bench_hbm_$TAG.json 2> $OUT/eval_bench_hbm_$TAG.err; tail -1 $OUT/eval_bench_hbm_$TAG.json | cut -c1-900 ;;
    tests_eval)
      timeout 600 python -m pytest tests/test_search_gpu.py -m gpu -q --timeout 300 -x -k "eval" > $OUT/pytest_eval_$TAG.log 2>&1; tail -5 $OUT/pytest_eval_$TAG.log ;;
    tests_attn)
      timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x -k "attn or attention or model" > $OUT/pytest_attn_$TAG.log 2>&1; tail -5 $OUT/pytest_attn_$TAG.log ;;
    bench_attn_ab)  # the attention model's serving rate: default forms (both on the pre-projected table) against the forms on the embedding rows
      timeout 300 python tools/attn_bench.py /tmp/idx 512 > $OUT/bench_attn_$TAG.txt 2> $OUT/bench_attn_$TAG.err
      NANN_PREPROJECT=0 timeout 300 python tools/attn_bench.py /tmp/idx 512 >> $OUT/bench_attn_$TAG.txt 2>> $OUT/bench_attn_$TAG.err
      cut -c1-600 $OUT/bench_attn_$TAG.txt; tail -2 $OUT/bench_attn_$TAG.err | grep -v amdgpu.ids ;;
    bench_mlp_small)  # where does the pipeline of phases (17 launches per chunk) stop paying?  batch sweep, mapping 6 against the fused kernel (5)
      for B in ${SMALL_BATCHES:-32 128 256 512 2048 4096}; do for M in 6 5; do
        NANN_MLP_MAPPING=$M timeout 200 $BENCH --scorer mlp --mlp-precision ${SMALL_PREC:-split} --batch $B --steps ${SMALL_STEPS:-60} --warmup ${SMALL_STEPS:-60} --no-secondary --no-cpu-baseline > $OUT/bench_mlp_small_b${B}_map${M}_$TAG.json 2> $OUT/bench_mlp_small_b${B}_map${M}_$TAG.err
        python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_mlp_small_b${B}_map${M}_$TAG.json').read().strip().splitlines()[-1])
    print('SMALL batch %5d mapping $M  qps %9.0f  ms/step %7.4f  kernel_ms %7.4f' % ($B, d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))
except Exception as e:
    print('SMALL $B $M failed', e)
PY
      done; done ;;
    bench_mlp_wide_ab)  # config 5's row shape and beam under the MLP (1.2M x 256-d bf16, ef=256), steady state: the pipeline of phases on the 32K-slot plan (default) against the fused HBM-bitmap kernel (NANN_MLP_MAPPING=5), both precisions
      for P in split exact; do for M in 6 5; do
        S="--items 1200000 --dim 256 --dtype bf16 --ef 256 --scorer mlp --mlp-precision $P --batch 1024 --steps 20 --warmup 30 --no-secondary --no-cpu-baseline"
        NANN_MLP_MAPPING=$M timeout 400 $BENCH $S > $OUT/bench_mlp_wide_${P}_map${M}_$TAG.json 2> $OUT/bench_mlp_wide_${P}_map${M}_$TAG.err
        show $OUT/bench_mlp_wide_${P}_map${M}_$TAG.json "MLP_WIDE_${P}_MAPPING_$M"; tail -2 $OUT/bench_mlp_wide_${P}_map${M}_$TAG.err | grep -v amdgpu.ids
      done; done ;;
    prof_default)  # rocprofv3 --kernel-trace --stats of the DEFAULT bench command (every workload of its JSON line), CPU legs shortened
      rm -rf /tmp/prof/kt_default
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt_default -o kt -- $BENCH --cpu-seconds 3 > $OUT/bench_profiled_$TAG.json 2> $OUT/prof_kt_default_$TAG.log )
      find /tmp/prof/kt_default -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_default_bench_$TAG.csv \;
      head -14 $OUT/kernel_stats_default_bench_$TAG.csv | cut -c1-200 ;;
    *) echo "unknown step $STEP" ;;
  esac
done
cat $OUT/pmc_$TAG.txt
echo "done left=$(left)"
