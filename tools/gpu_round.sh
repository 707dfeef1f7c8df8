#!/bin/bash
# One call on the MI355X box, most important first, each step skipped once DEADLINE seconds
# have passed: GPU tests, smoke, the bench.py line (+CPU baseline, secondary workloads), traversal-mode
# A/B on the same cached index, rocprofv3 kernel stats, HBM PMC passes.
# usage: tools/gpu_round.sh <tag> [deadline_s] [modes: "lds_hash lds_bitmap ..."] [extra bench flags]
set -u
TAG=${1:-final}
DEADLINE=${2:-600}
MODES=${3:-"lds_bitmap"}
EXTRA=${4:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
T0=$(date +%s)
left() { echo $(( DEADLINE - ($(date +%s) - T0) )); }
mkdir -p $OUT /tmp/idx /tmp/prof
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_$TAG.log 2>&1
RC=$?; tail -3 $OUT/pytest_$TAG.log
if [ $RC -ne 0 ]; then echo "GPU TESTS FAILED"; grep -E "^(E  |FAILED|ERROR)" $OUT/pytest_$TAG.log | head -30; fi
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; tail -1 $OUT/smoke_$TAG.log
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --index-cache /tmp/idx $EXTRA"
show() {  # file label
python - <<PY
import json
try:
    d = json.loads(open('$1').read().strip().splitlines()[-1])
    r = d['roofline']
    print('$2', 'qps', d['value'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'trav', d.get('traversal'),
          'cpu', d.get('cpu_baseline', {}).get('value'), 'parity', d.get('parity'), 'recall', d.get('recall_at_k_vs_bruteforce'),
          'setup_s', d.get('setup_s'), 'deg0', d.get('mean_degree_l0'), 'rows/q', r.get('rows_scored_per_query'))
    if 'phase_breakdown' in d:
        print('   ticks', {k: round(v) for k, v in d['phase_breakdown']['ticks_per_query'].items()})
    for k, v in d.get('secondary', {}).items():
        if isinstance(v, dict) and 'roofline' in v:
            print('   SEC', k, 'qps', v['qps_end_to_end'], 'kernel_ms', v['roofline']['kernel_ms'], 'frac', v['roofline']['frac'],
                  'trav', v.get('traversal'), 'parity', v.get('parity'), 'setup_s', v.get('setup_s'))
        else:
            print('   SEC', k, v)
except Exception as e:
    print('$2 parse failed', e)
PY
}
timeout 420 $BENCH --phase-ticks --cpu-seconds 8 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$? left=$(left)"
show $OUT/bench_$TAG.json DEFAULT
tail -3 $OUT/bench_$TAG.err
for M in $MODES; do
  if [ $(left) -gt 60 ]; then
    timeout 120 $BENCH --traversal $M --phase-ticks --no-cpu-baseline --no-secondary --steps 10 > $OUT/bench_${TAG}_$M.json 2> $OUT/bench_${TAG}_$M.err
    show $OUT/bench_${TAG}_$M.json "MODE $M"
  fi
done
for d in $R/nann_amd/_build/var_*; do  # tools/build_variants.py
  [ -f "$d/libnann_hip.so" ] || continue
  V=$(basename $d | sed 's/^var_//')
  if [ $(left) -gt 60 ]; then
    NANN_HIP_LIB=$d/libnann_hip.so timeout 120 $BENCH --phase-ticks --no-cpu-baseline --no-secondary --steps 10 > $OUT/bench_${TAG}_var_$V.json 2> $OUT/bench_${TAG}_var_$V.err
    show $OUT/bench_${TAG}_var_$V.json "VARIANT $V"
  fi
done
if [ $(left) -gt 60 ]; then
  rm -rf /tmp/prof/kt
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- \
      $BENCH --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/prof_kt_$TAG.log 2>&1
  find /tmp/prof/kt -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$TAG.csv \;
  grep -E "k_search" $OUT/kernel_stats_$TAG.csv | head -3
fi
: > $OUT/pmc_$TAG.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  if [ $(left) -gt 50 ]; then
    N=$(echo $C | cut -d' ' -f1)
    rm -rf /tmp/prof/pmc_$N
    timeout 120 rocprofv3 --pmc $C --output-format csv -d /tmp/prof/pmc_$N -o pmc -- \
        $BENCH --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/prof_pmc_${N}_$TAG.log 2>&1
    python - <<PY >> $OUT/pmc_$TAG.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/prof/pmc_$N/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_search' in r.get('Kernel_Name', ''):
            acc[(r['Kernel_Name'][:60], r.get('Counter_Name'))].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    print(c, k, 'dispatches', len(v), 'mean', sum(v) / len(v), 'min', min(v), 'max', max(v))
PY
  fi
done
cat $OUT/pmc_$TAG.txt
echo "done left=$(left)"
