cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_p1
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-latency-line --steps 100 --warmup 10 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
echo rc=$?
ls $OUT/kt | head
python3 - <<'PY'
import csv,os,glob
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03_p1'
f=glob.glob(out+'/kt/*kernel_stats.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if not r['Name'].startswith(('void at::', 'void rocblas', 'at::'))]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]: print('%-60s %7s calls  avg %9.1f us  %5.1f%%'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
print('total kernel ms', tot/1e6)
# busy time: union of kernel intervals
t=glob.glob(out+'/kt/*kernel_trace.csv')[0]
iv=[]
for r in csv.DictReader(open(t)):
    if r['Kernel_Name'].startswith(('void at::', 'void rocblas', 'at::')): continue      # (the set-up's renderer)
    iv.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
iv.sort()
busy=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print('span ms', (iv[-1][1]-iv[0][0])/1e6, 'busy (union) ms', busy/1e6, 'sum ms', sum(e-s for s,e in iv)/1e6)
PY
cp $OUT/kt/*kernel_stats.csv $OUT/kernel_stats.csv; rm -rf $OUT/kt
