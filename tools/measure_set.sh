#!/bin/bash
# One measurement set for profiles/: bench lines, kernel stats, PMC passes.  usage (on the GPU box, repo root): tools/measure_set.sh r02_a
set -o pipefail
TAG=${1:-r02_x}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
PROF_SHAPE="${PROF_SHAPE:-}"                                   # kernel trace: the default command (two lockstep groups)
PMC_SHAPE="${PMC_SHAPE:---sessions 96 --lanes 96}"            # counter passes: ONE group (rocprofv3 --pmc segfaults with two groups' launching threads)
if [ -z "$SKIP_BENCH" ]; then
python3 bench.py > $OUT/${TAG}_c2_bench.json 2> $OUT/c2.err || exit 1
echo "c2 bench done"; cut -c1-160 $OUT/${TAG}_c2_bench.json
python3 bench.py --config c3 > $OUT/${TAG}_c3_bench.json 2> $OUT/c3.err || exit 1
echo "c3 bench done"
python3 bench.py --host-images --no-cpu-baseline --no-latency-line > $OUT/${TAG}_c2_bench_host_images.json 2> $OUT/hi.err || exit 1
echo "host-images bench done"
python3 bench.py --no-cpu-baseline --no-latency-line --steps 100 --sweep 1x0,1x1,4x4,16x16,32x32,64x64,128x64,192x64,192x96,256x128 > $OUT/${TAG}_c2_sweep.json 2> $OUT/sw.err || exit 1
echo "sweep done"
# what the hand-over delay costs: 2 and 8 frames instead of 4
python3 bench.py --no-cpu-baseline --no-latency-line --mapping-delay 2 > $OUT/${TAG}_c2_bench_delay2.json 2> $OUT/delay2.err || exit 1
python3 bench.py --no-cpu-baseline --no-latency-line --mapping-delay 8 > $OUT/${TAG}_c2_bench_delay8.json 2> $OUT/delay8.err || exit 1
echo "lag variants done"
python3 tools/extract_rate.py 128 20 > $OUT/${TAG}_extract_rate.txt 2> $OUT/er.err || exit 1
echo "extract rate done"
fi
cd /tmp && export TMPDIR=/tmp
if [ -z "$SKIP_TRACE" ]; then
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/bench.py --no-cpu-baseline --no-latency-line --steps 100 --warmup 10 $PROF_SHAPE > $OUT/${TAG}_c2_bench_under_rocprof.json 2> $OUT/kt.err || exit 1
cp $OUT/kt/kt_kernel_stats.csv $OUT/${TAG}_c2_kernel_stats.csv
echo "kernel trace done"
fi
if [ -n "$SKIP_PMC" ]; then exit 0; fi
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVES"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o p -- python3 $REPO/bench.py --no-cpu-baseline --no-latency-line --steps 60 --warmup 10 $PMC_SHAPE > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err || exit 1
  echo "pmc $N done"
done
cd $REPO
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_summary.json $OUT/pmc_*/p_counter_collection.csv
