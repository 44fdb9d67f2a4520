#!/bin/bash
# Reduced measurement set of a final build: default line, the driver's steps, two-group kernel trace, PMC passes.  usage: tools/measure_final.sh r03_c
set -o pipefail
TAG=${1:-r03_x}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
python3 bench.py > $OUT/${TAG}_c2_bench.json 2> $OUT/c2.err || exit 1
echo "c2 bench done"; cut -c1-140 $OUT/${TAG}_c2_bench.json
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_c2_bench_steps20.json 2> $OUT/c2s.err || exit 1
echo "steps20 done"; cut -c1-140 $OUT/${TAG}_c2_bench_steps20.json
python3 bench.py --config c3 --no-cpu-baseline --no-latency-line > $OUT/${TAG}_c3_bench.json 2> $OUT/c3.err || exit 1
echo "c3 done"; cut -c1-140 $OUT/${TAG}_c3_bench.json
cd /tmp && export TMPDIR=/tmp
export VSLAM_DUMP_MAPS=$OUT/maps.txt
for attempt in 1 2; do
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/bench.py --no-cpu-baseline --no-latency-line --steps 100 --warmup 10 > $OUT/${TAG}_c2_bench_under_rocprof.json 2> $OUT/kt.err
  RC=$?
  if grep -q "SIGSEGV\|Aborted at" $OUT/kt.err; then
    cp $OUT/kt.err $OUT/kt_crash_$attempt.err
    python3 $REPO/tools/resolve_stack.py $OUT/maps.txt $OUT/kt.err > $OUT/kt_crash_${attempt}_resolved.txt 2>&1
    echo "kernel trace attempt $attempt crashed (rc $RC)"; head -12 $OUT/kt_crash_${attempt}_resolved.txt
    rm -rf $OUT/kt; continue
  fi
  [ $RC -eq 0 ] || exit 1
  break
done
unset VSLAM_DUMP_MAPS
cp $OUT/kt/kt_kernel_stats.csv $OUT/${TAG}_c2_kernel_stats.csv || exit 1
rm -rf $OUT/kt
echo "kernel trace done"
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVES"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o p -- python3 $REPO/bench.py --no-cpu-baseline --no-latency-line --steps 60 --warmup 10 --sessions 128 --lanes 128 > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err || exit 1
  echo "pmc $N done"
done
cd $REPO
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_summary.json --meta '{"config": "c2", "lanes": 128, "scene": "corridor", "shape": "--sessions 128 --lanes 128"}' $OUT/pmc_*/p_counter_collection.csv
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU_MFMA_MOPS_F64
# C5 (64 keyframes / 100 k landmarks): bench line + kernel trace of the same command
python3 bench.py --config c5 > $OUT/${TAG}_c5_bench.json 2> $OUT/c5.err || exit 1
echo "c5 done"; cut -c1-200 $OUT/${TAG}_c5_bench.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt5 -o kt -- python3 $REPO/bench.py --config c5 > $OUT/c5_under_rocprof.json 2> $OUT/kt5.err || exit 1
cp $OUT/kt5/kt_kernel_stats.csv $OUT/${TAG}_c5_kernel_stats.csv; rm -rf $OUT/kt5
cd $REPO
echo "all done"
