#!/bin/bash
# true kernel durations of a local-BA cohort alone (rocprofv3 kernel trace): tools/prof_cohort.sh <tag> [ENV=..] ...
TAG=$1; shift
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export $e; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/tools/bacohort.py 10 > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/kt/*kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'vslam' in r['Name']]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:12]: print('%-56s %6s calls avg %8.1f us min %7.1f max %7.1f'%(r['Name'][:56], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
rm -rf $OUT/kt
