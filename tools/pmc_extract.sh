#!/bin/bash
# PMC counters of the extraction kernels alone (192 images of the bench's corridor scene): tools/pmc_extract.sh
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmc_extract
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_WAVE32_LDS SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o p -- python3 $REPO/tools/extract_rate.py 192 4 corridor > $OUT/out$i.txt 2> $OUT/err$i.txt || { echo "pass $i failed"; tail -3 $OUT/err$i.txt; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('vslam::','').split('(')[0].replace('void ','')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    if not k.startswith('k_'): continue
    print(k, {c: round(sum(v)/len(v)) for c,v in sorted(agg[k].items())})
PY
cat $OUT/out1.txt | tail -1
rm -rf $OUT/p*
