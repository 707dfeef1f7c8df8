#!/bin/bash
# One-call verification + measurement on the MI355X box: GPU tests, smoke, the bench.py line,
# rocprofv3 kernel stats and HBM-traffic PMC passes for the L2 workload, then the MLP workload.
# Stops after the tests if they fail.  usage: tools/gpu_final.sh <tag>
set -u
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT /tmp/idx /tmp/prof
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -x > $OUT/pytest_$TAG.log 2>&1
RC=$?; tail -4 $OUT/pytest_$TAG.log
if [ $RC -ne 0 ]; then echo "GPU TESTS FAILED"; grep -E "^(E  |FAILED)" $OUT/pytest_$TAG.log | head -20; exit 1; fi
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; tail -1 $OUT/smoke_$TAG.log
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --index-cache /tmp/idx"
timeout 500 $BENCH --phase-ticks > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
tail -c 2400 $OUT/bench_$TAG.json
rm -rf /tmp/prof/kt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- \
    $BENCH --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_kt_$TAG.log 2>&1
find /tmp/prof/kt -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$TAG.csv \;
grep -E "k_search|k_user_seq" $OUT/kernel_stats_$TAG.csv | head -3
: > $OUT/pmc_$TAG.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof/pmc_$C
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/prof/pmc_$C -o pmc -- \
      $BENCH --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_${C}_$TAG.log 2>&1
  python - <<PY >> $OUT/pmc_$TAG.txt 2>&1
import csv, glob
for f in glob.glob('/tmp/prof/pmc_$C/**/*counter_collection.csv', recursive=True):
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
            if 'k_search' in r.get('Kernel_Name', '') and r.get('Counter_Name') == '$C']
    if vals:
        print('$C k_search dispatches', len(vals), 'mean', sum(vals) / len(vals), 'min', min(vals), 'max', max(vals))
PY
done
cat $OUT/pmc_$TAG.txt
# ---- MLP scorer (BASELINE configs[2])
MB="$BENCH --scorer mlp --batch 512 --steps 3 --warmup 1 --cpu-seconds 6"
timeout 300 $MB > $OUT/bench_${TAG}_mlp.json 2> $OUT/bench_${TAG}_mlp.err; echo "mlp bench rc=$?"
tail -c 1800 $OUT/bench_${TAG}_mlp.json
rm -rf /tmp/prof/ktm
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/ktm -o kt -- \
    $MB --steps 2 --no-cpu-baseline > $OUT/prof_kt_${TAG}_mlp.log 2>&1
find /tmp/prof/ktm -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_${TAG}_mlp.csv \;
grep k_search $OUT/kernel_stats_${TAG}_mlp.csv | head -2
rm -rf /tmp/prof/pmcm
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof/pmcm -o pmc -- \
    $MB --steps 2 --no-cpu-baseline > $OUT/prof_pmc_${TAG}_mlp.log 2>&1
python - <<PY > $OUT/pmc_${TAG}_mlp.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/prof/pmcm/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_search' in r.get('Kernel_Name', ''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k, 'k_search dispatches', len(v), 'mean', sum(v) / len(v))
PY
cat $OUT/pmc_${TAG}_mlp.txt
