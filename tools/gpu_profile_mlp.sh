#!/bin/bash
# MLP-scorer run (BASELINE configs[2]): bench + kernel stats + MFMA-busy PMC pass.
set -u
TAG=${1:-r1mlp}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT /tmp/idx /tmp/prof
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --index-cache /tmp/idx --scorer mlp --batch 1024 --steps 5 --warmup 1 --cpu-seconds 10 $*"
timeout 1200 $BENCH > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?" >> $OUT/bench_$TAG.err
rm -rf /tmp/prof/kt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- \
    $BENCH --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_kt_$TAG.log 2>&1
find /tmp/prof/kt -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$TAG.csv \;
rm -rf /tmp/prof/pmc
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof/pmc -o pmc -- \
    $BENCH --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_$TAG.log 2>&1
python - <<PY > $OUT/pmc_$TAG.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/prof/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_search' in r.get('Kernel_Name', ''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k, 'k_search dispatches', len(v), 'mean', sum(v) / len(v))
PY
cat $OUT/pmc_$TAG.txt; tail -c 2000 $OUT/bench_$TAG.json; tail -2 $OUT/bench_$TAG.err; grep k_search $OUT/kernel_stats_$TAG.csv | head -3
