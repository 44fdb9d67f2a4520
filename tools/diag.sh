cd /root/repo
for v in "--scene room" "--scene corridor --sessions 8 --lanes 8 --mapping 1" "--scene corridor --sessions 8 --lanes 0 --mapping 1" "--scene corridor --sessions 8 --lanes 8 --mapping 0" "--scene corridor --sessions 8 --lanes 8 --mapping 2"; do
  echo "== $v"
  python bench.py --no-cpu-baseline --no-latency-line --steps 100 $v 2> gpurun_out/diag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), d['tracking']); print(d['config']['workload'][-260:])"
done
