#!/bin/bash
# throughput of a few host-side shapes (each argument: "bench args|ENV=..,ENV=..") with the host phase lines
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  env VSLAM_BATCH_PHASES=1 "${@:2}" python bench.py --no-cpu-baseline --no-latency-line --steps 100 $1 2> gpurun_out/hv.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'lost', d['tracking']['lost_frames'], 'BAs', d['tracking']['local_bas'], 'rms', round(d['tracking']['rms_position_error_m'],4))
print({k: round(v*1e3,2) for k,v in d['stage_ms_per_frame'].items() if k.startswith('ba_')})"
  grep -E "host phases|begin =|cohorts|fetch_keys|batch host|ba_collect parts|sections" gpurun_out/hv.err | cut -c1-420 | head -8
}
for v in "$@"; do
  IFS='|' read -r a e <<< "$v"
  run "$a" $(echo $e | tr ',' ' ')
done
