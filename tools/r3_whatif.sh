#!/bin/bash
# On the GPU box: bench.py at the driver's flags under a list of environment settings (timing-only what-if runs), one line each.
#   tools/r3_whatif.sh <tag> "ENV1=a ENV2=b" "ENV3=c" ...     ("-" = no extra environment)
tag=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  out=$(env $e timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io-steps 0 2>>gpurun_out/${tag}.err | tail -1)
  echo "$out" | python3 -c "
import json,sys
e=sys.argv[1]
try:
    d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_call']
    print('%-60s %8.1f Mpts/s  stages %s  normals %.1f  lat %.1f ms  mism %s' % (e or '(default)', d['value'] or -1, ' '.join('%.0f' % v for v in s.values()), d['roofline']['stages']['normals kernel (inside neighbours+normals)']['ms_per_call'], d['single_frame_latency_ms'], (d.get('labels_checked') or {}).get('mismatches')))
except Exception as ex: print(e, 'failed', ex)
" "$e" | tee -a gpurun_out/${tag}.log
done
