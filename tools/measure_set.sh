#!/bin/bash
# One measurement set for profiles/: bench lines, kernel stats, PMC passes.  usage (on the GPU box, repo root): tools/measure_set.sh r03_a
set -o pipefail
TAG=${1:-r03_x}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
PROF_SHAPE="${PROF_SHAPE:-}"                                   # kernel trace: the default command (two lockstep groups)
PMC_SHAPE="${PMC_SHAPE:---sessions 128 --lanes 128}"            # counter passes: ONE group (the default shape's launch geometry per kernel)
if [ -z "$SKIP_BENCH" ]; then
python3 bench.py > $OUT/${TAG}_c2_bench.json 2> $OUT/c2.err || exit 1
echo "c2 bench done"; cut -c1-160 $OUT/${TAG}_c2_bench.json
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_c2_bench_steps20.json 2> $OUT/c2s.err || exit 1
echo "c2 bench (driver's steps) done"; cut -c1-160 $OUT/${TAG}_c2_bench_steps20.json
python3 bench.py --config c3 > $OUT/${TAG}_c3_bench.json 2> $OUT/c3.err || exit 1
echo "c3 bench done"; cut -c1-160 $OUT/${TAG}_c3_bench.json
python3 bench.py --host-images --no-cpu-baseline --no-latency-line > $OUT/${TAG}_c2_bench_host_images.json 2> $OUT/hi.err || exit 1
echo "host-images bench done"
python3 bench.py --no-cpu-baseline --no-latency-line --steps 100 --sweep 1x0,1x1,16x16,64x64,96x96,192x64,192x96,256x128 > $OUT/${TAG}_c2_sweep.json 2> $OUT/sw.err || exit 1
echo "sweep done"
# what the hand-over delay costs: 2 and 8 frames instead of 4
python3 bench.py --no-cpu-baseline --no-latency-line --mapping-delay 2 > $OUT/${TAG}_c2_bench_delay2.json 2> $OUT/delay2.err || exit 1
python3 bench.py --no-cpu-baseline --no-latency-line --mapping-delay 8 > $OUT/${TAG}_c2_bench_delay8.json 2> $OUT/delay8.err || exit 1
echo "delay variants done"
python3 bench.py --no-cpu-baseline --no-latency-line --scene room > $OUT/${TAG}_c2_bench_room.json 2> $OUT/room.err || exit 1
echo "room scene (round 2's workload) done"; cut -c1-160 $OUT/${TAG}_c2_bench_room.json
python3 tools/extract_rate.py 128 20 > $OUT/${TAG}_extract_rate.txt 2> $OUT/er.err || exit 1
echo "extract rate done"
python3 bench.py --config c5 > $OUT/${TAG}_c5_bench.json 2> $OUT/c5.err || exit 1
echo "c5 bench done"; cut -c1-200 $OUT/${TAG}_c5_bench.json
fi
cd /tmp && export TMPDIR=/tmp
if [ -z "$SKIP_TRACE" ]; then
# the process keeps a copy of its address map on disk: a crash trace of the profiled run can be resolved (tools/resolve_stack.py)
export VSLAM_DUMP_MAPS=$OUT/maps.txt
for attempt in 1 2; do
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/bench.py --no-cpu-baseline --no-latency-line --steps 100 --warmup 10 $PROF_SHAPE > $OUT/${TAG}_c2_bench_under_rocprof.json 2> $OUT/kt.err
  RC=$?
  if grep -q "SIGSEGV\|Aborted at" $OUT/kt.err; then
    cp $OUT/kt.err $OUT/kt_crash_$attempt.err; cp $OUT/maps.txt $OUT/kt_crash_${attempt}_maps.txt
    python3 $REPO/tools/resolve_stack.py $OUT/maps.txt $OUT/kt.err > $OUT/kt_crash_${attempt}_resolved.txt 2>&1
    echo "kernel trace attempt $attempt crashed (rc $RC); resolved trace:"; cat $OUT/kt_crash_${attempt}_resolved.txt
    rm -rf $OUT/kt
    continue
  fi
  [ $RC -eq 0 ] || exit 1
  break
done
unset VSLAM_DUMP_MAPS
cp $OUT/kt/kt_kernel_stats.csv $OUT/${TAG}_c2_kernel_stats.csv || exit 1
rm -rf $OUT/kt
echo "kernel trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt5 -o kt -- python3 $REPO/bench.py --config c5 --c5-steps 3 > $OUT/${TAG}_c5_bench_under_rocprof.json 2> $OUT/kt5.err || exit 1
cp $OUT/kt5/kt_kernel_stats.csv $OUT/${TAG}_c5_kernel_stats.csv; rm -rf $OUT/kt5
echo "c5 kernel trace done"
fi
if [ -n "$SKIP_PMC" ]; then exit 0; fi
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVES"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o p -- python3 $REPO/bench.py --no-cpu-baseline --no-latency-line --steps 60 --warmup 10 $PMC_SHAPE > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err || exit 1
  echo "pmc $N done"
done
cd $REPO
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_summary.json --meta '{"config": "c2", "lanes": 128, "scene": "corridor", "shape": "'"$PMC_SHAPE"'"}' $OUT/pmc_*/p_counter_collection.csv
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU_MFMA_MOPS_F64
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVES"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc5_$N -o p -- python3 $REPO/bench.py --config c5 --c5-steps 2 > $OUT/pmc5_$N.json 2> $OUT/pmc5_$N.err || exit 1
  echo "c5 pmc $N done"
done
cd $REPO
python3 tools/pmc_summary.py $OUT/${TAG}_c5_pmc_summary.json $OUT/pmc5_*/p_counter_collection.csv
rm -rf $OUT/pmc5_*
