#!/bin/bash
# PMC counters of the local-BA cohort kernels alone: tools/pmc_cohort.sh <kernel substring> [ENV=..]
K=$1; shift
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmc_cohort
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export $e; done
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o p -- python3 $REPO/tools/bacohort.py 10 > $OUT/out$i.txt 2> $OUT/err$i.txt || { echo "pass $i failed"; tail -3 $OUT/err$i.txt; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$K" in r['Kernel_Name']:
            agg[r['Counter_Name']][r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
for c in sorted(agg):
    for k,v in agg[c].items():
        v=sorted(v); print('%-28s %-40s n=%d mean %12.0f  p90 %12.0f max %12.0f'%(c,k,len(v),sum(v)/len(v),v[int(0.9*len(v))],v[-1]))
PY
rm -rf $OUT/p*
