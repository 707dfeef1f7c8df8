#!/bin/bash
# One call on the MI355X box, most important first, each step skipped once DEADLINE seconds
# have passed: GPU tests, smoke, bench.py line (+CPU baseline), rocprofv3 kernel stats, HBM
# PMC passes, kernel-variant A/B runs (prebuilt libs under nann_amd/_build/var_*), MLP bench.
# usage: tools/gpu_round.sh <tag> [deadline_s] [variants: yes|no] [glb]
set -u
TAG=${1:-final}
DEADLINE=${2:-320}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
T0=$(date +%s)
left() { echo $(( DEADLINE - ($(date +%s) - T0) )); }
mkdir -p $OUT /tmp/idx /tmp/prof
cd $R
timeout 200 python -m pytest tests -m gpu -q --timeout 120 -x > $OUT/pytest_$TAG.log 2>&1
RC=$?; tail -3 $OUT/pytest_$TAG.log
if [ $RC -ne 0 ]; then echo "GPU TESTS FAILED"; grep -E "^(E  |FAILED)" $OUT/pytest_$TAG.log | head -20; exit 1; fi
# kernels written without hardware access run here first, outside the gate above
NANN_RUN_UNVERIFIED=1 timeout 120 python -m pytest tests -m gpu -q --timeout 100 -k "attn_scorer" > $OUT/pytest_unverified_$TAG.log 2>&1
echo "unverified tests rc=$?"; tail -3 $OUT/pytest_unverified_$TAG.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; tail -1 $OUT/smoke_$TAG.log
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --index-cache /tmp/idx"
timeout 260 $BENCH --phase-ticks --cpu-seconds 8 --batch-sweep 1,64,1024 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$? left=$(left)"
python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$TAG.json').read().strip().splitlines()[-1])
    print('DEFAULT', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('cpu_baseline', {}).get('value'), d.get('parity'))
    print({k: round(v) for k, v in d['phase_breakdown']['ticks_per_query'].items()})
except Exception as e:
    print('bench parse failed', e)
PY
if [ $(left) -gt 45 ]; then
  rm -rf /tmp/prof/kt
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- \
      $BENCH --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_kt_$TAG.log 2>&1
  find /tmp/prof/kt -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$TAG.csv \;
  grep -E "k_search" $OUT/kernel_stats_$TAG.csv | head -2
fi
: > $OUT/pmc_$TAG.txt
for C in FETCH_SIZE WRITE_SIZE; do
  if [ $(left) -gt 40 ]; then
    rm -rf /tmp/prof/pmc_$C
    timeout 100 rocprofv3 --pmc $C --output-format csv -d /tmp/prof/pmc_$C -o pmc -- \
        $BENCH --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_${C}_$TAG.log 2>&1
    python - <<PY >> $OUT/pmc_$TAG.txt 2>&1
import csv, glob
for f in glob.glob('/tmp/prof/pmc_$C/**/*counter_collection.csv', recursive=True):
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
            if 'k_search' in r.get('Kernel_Name', '') and r.get('Counter_Name') == '$C']
    if vals:
        print('$C k_search dispatches', len(vals), 'mean', sum(vals) / len(vals), 'min', min(vals), 'max', max(vals))
PY
  fi
done
cat $OUT/pmc_$TAG.txt
# ---- what the memory system delivers for the scoring phase's access pattern
if [ -x $R/tools/_build/ubench_gather ] && [ $(left) -gt 20 ]; then
  timeout 60 $R/tools/_build/ubench_gather > $OUT/ubench_gather_$TAG.txt 2>&1; cat $OUT/ubench_gather_$TAG.txt
fi
# ---- kernel variants (same index, no CPU baseline)
VARIANTS=${3:-yes}
run_variant() {  # name, env assignments...
  local name=$1; shift
  for kv in "$@"; do  # a variant whose prebuilt library is not there is skipped
    case $kv in NANN_HIP_LIB=*) [ -f "${kv#NANN_HIP_LIB=}" ] || return 0;; esac
  done
  [ "$VARIANTS" = yes ] || return 0
  if [ $(left) -gt 30 ]; then
    env "$@" timeout 90 $BENCH --phase-ticks --no-cpu-baseline --steps 10 > $OUT/bench_${TAG}_$name.json 2> $OUT/bench_${TAG}_$name.err
    python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_${TAG}_$name.json').read().strip().splitlines()[-1])
    t = d['phase_breakdown']['ticks_per_query']
    print('VARIANT $name', d['value'], d['roofline']['kernel_ms'], d.get('parity', {}).get('ids_equal'), 'expand', round(t['expand']), 'score', round(t['score']), 'topk', round(t['topk']), 'filter', round(t['ex_walkbusy']))
except Exception as e:
    print('variant $name failed', e)
PY
  fi
}
for d in $R/nann_amd/_build/var_*; do  # tools/build_variants.py
  [ -d "$d" ] && run_variant $(basename $d | sed 's/^var_//') NANN_HIP_LIB=$d/libnann_hip.so
done
# ---- MLP scorer (BASELINE configs[2])
if [ $(left) -gt 35 ]; then
  MB="$BENCH --scorer mlp --batch 512 --steps 3 --warmup 1 --no-cpu-baseline"
  timeout 100 $MB > $OUT/bench_${TAG}_mlp.json 2> $OUT/bench_${TAG}_mlp.err; echo "mlp bench rc=$?"
  tail -c 900 $OUT/bench_${TAG}_mlp.json
fi
[ "${4:-}" = glb ] && run_variant glb512 NANN_L2_VARIANT=glb512
echo "done left=$(left)"
