#!/bin/bash
# rocprofv3 PMC passes (counters only, one pass per group) over an arbitrary command on the MI355X box;
# per-kernel means of every counter go to gpurun_out/pmc_<tag>.txt.
# usage: tools/gpu_pmc_cmd.sh <tag> <kernel-name substring> <command...>
set -u
TAG=$1; KSUB=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT /tmp/prof
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc_$TAG.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/prof/p$i
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/prof/p$i -o pmc -- "$@" > $OUT/prof_pmc_${TAG}_$i.log 2>&1
  python - <<PY >> $OUT/pmc_$TAG.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/prof/p$i/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if '$KSUB' in r.get('Kernel_Name', ''):
            acc[(r['Kernel_Name'][:70], r.get('Counter_Name'))].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    print(c, k, 'dispatches', len(v), 'mean', sum(v) / len(v))
PY
done
cat $OUT/pmc_$TAG.txt
