#!/bin/bash
# A/B of library variants (nann_amd/_build/var_*/, tools/build_variants.py) on the GPU box:
# the same bench.py command for the default library and for every variant, one index build.
# usage: tools/gpu_ab.sh <tag> "<bench flags>"
set -u
TAG=${1:-ab}
FLAGS=${2:-"--no-cpu-baseline --no-secondary --steps 10"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT /tmp/idx
cd /tmp && export TMPDIR=/tmp
run() {  # name [env...]
  local name=$1; shift
  env "$@" timeout 300 python $R/bench.py --index-cache /tmp/idx $FLAGS > $OUT/ab_${TAG}_$name.json 2> $OUT/ab_${TAG}_$name.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/ab_${TAG}_$name.json').read().strip().splitlines()[-1])
    r = d['roofline']
    print('$name', 'qps', d['value'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'parity', d.get('parity'))
    if 'phase_breakdown' in d:
        print('   ticks', {k: round(v) for k, v in d['phase_breakdown']['ticks_per_query'].items()})
except Exception as e:
    print('$name failed', e); print(open('$OUT/ab_${TAG}_$name.err').read()[-1500:])
PY
}
run default A=1
for d in $R/nann_amd/_build/var_*; do
  [ -f "$d/libnann_hip.so" ] || continue
  run $(basename $d | sed 's/^var_//') NANN_HIP_LIB=$d/libnann_hip.so
done
