#!/bin/bash
# Runs on the MI355X box (through gpurun): bench + rocprofv3 kernel stats + PMC HBM traffic.
# Summaries land in gpurun_out/ (copy the ones to keep into profiles/).
# usage: tools/gpu_profile.sh <tag> [extra bench args]
set -u
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT /tmp/idx /tmp/prof
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --index-cache /tmp/idx $*"

# 1. plain bench (builds + caches the index), with per-phase attribution
timeout 900 $BENCH --phase-ticks > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?" >> $OUT/bench_$TAG.err

# 2. kernel trace + stats of the same command (fewer steps; no CPU baseline)
rm -rf /tmp/prof/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o kt -- \
    $BENCH --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_kt_$TAG.log 2>&1
find /tmp/prof/kt -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$TAG.csv \;
python - <<PY >> $OUT/prof_kt_$TAG.log 2>&1
import csv, glob
for f in glob.glob('/tmp/prof/kt/**/*kernel_trace.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'k_search' in r.get('Kernel_Name', '')]
    if rows:
        d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
        print('k_search dispatches', len(d), 'avg_ms', sum(d) / len(d), 'min_ms', min(d), 'max_ms', max(d),
              'vgpr', rows[0].get('VGPR_Count'), 'sgpr', rows[0].get('SGPR_Count'),
              'lds', rows[0].get('LDS_Block_Size'), 'scratch', rows[0].get('Scratch_Size'),
              'grid', rows[0].get('Grid_Size'), 'wg', rows[0].get('Workgroup_Size'))
PY

# 3. PMC passes (own runs, counters only): HBM bytes of k_search
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof/pmc_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/prof/pmc_$C -o pmc -- \
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
tail -3 $OUT/prof_kt_$TAG.log
tail -c 2500 $OUT/bench_$TAG.json
