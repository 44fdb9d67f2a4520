#!/bin/bash
# Kernel trace of the default command with the process's address map kept on disk, then the crash trace (if any) resolved.
# usage (GPU box, repo root): tools/crash_probe.sh <tag> [bench args]
TAG=${1:-probe}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VSLAM_DUMP_MAPS=$OUT/maps.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/bench.py --no-cpu-baseline --no-latency-line --steps 100 --warmup 10 "$@" > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
RC=$?
echo "rocprofv3 rc=$RC"
unset VSLAM_DUMP_MAPS
if grep -q "SIGSEGV\|Aborted at" $OUT/kt.err; then
  python3 $REPO/tools/resolve_stack.py $OUT/maps.txt $OUT/kt.err > $OUT/resolved.txt 2>&1
  cat $OUT/resolved.txt
fi
if ls $OUT/kt/*kernel_stats.csv > /dev/null 2>&1; then cp $OUT/kt/*kernel_stats.csv $OUT/kernel_stats.csv; fi
rm -rf $OUT/kt
exit 0
