#!/bin/bash
# On the GPU box: one call of 192 frames at a time (no other call in flight) under a list of environment settings: per-stage device time of a call.
cd "$GRAFT_REPO_ROOT" || exit 1
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  out=$(env $e timeout 400 python3 bench.py --groups 1 --batch 192 --steps 9 --warmup 3 --no-cpu-baseline --host-io-steps 0 2>/dev/null | tail -1)
  echo "$out" | python3 -c "
import json,sys
e=sys.argv[1]
d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_call']
print('%-40s %8.1f Mpts/s  stages %s  normals %.1f  mism %s' % (e or '(default)', d['value'] or -1, ' '.join('%.1f' % v for v in s.values()), d['roofline']['stages']['normals kernel (inside neighbours+normals)']['ms_per_call'], (d.get('labels_checked') or {}).get('mismatches')))
" "$e"
done
