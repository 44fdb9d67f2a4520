#!/bin/bash
# HBM traffic of the extraction kernels alone (192 corridor images): FETCH_SIZE / WRITE_SIZE passes (KB per launch; gfx950: 2 x FETCH)
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmc_extract_fetch
mkdir -p $OUT
python3 $REPO/tools/extract_rate.py 192 10 corridor 2>/dev/null | tail -1 > $OUT/extract_rate.txt
python3 $REPO/tools/extract_rate.py 192 10 room 2>/dev/null | tail -1 >> $OUT/extract_rate.txt
cat $OUT/extract_rate.txt
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p_$C -o p -- python3 $REPO/tools/extract_rate.py 192 4 corridor > $OUT/out_$C.txt 2> $OUT/err_$C.txt || { echo "pass $C failed"; tail -3 $OUT/err_$C.txt; }
done
python3 - <<PY > $OUT/extract_fetch.txt
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('vslam::','').split('(')[0].replace('void ','')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    if not k.startswith('k_'): continue
    print(k, {c: round(sum(v)/len(v)) for c,v in sorted(agg[k].items())}, "KB per launch (192 corridor images)")
PY
cat $OUT/extract_fetch.txt
rm -rf $OUT/p_*
