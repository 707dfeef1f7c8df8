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
    *) bash $R/tools/gpu_r5.sh $TAG $(left) $STEP ;;
  esac
done
echo "=== done in $(( $(date +%s) - T0 )) s"
