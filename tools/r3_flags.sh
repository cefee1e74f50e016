#!/bin/bash
# On the GPU box: bench.py at the driver's flags with extra command-line flags per run (one line each).   tools/r3_flags.sh <tag> "--groups 8 --batch 144" ...
tag=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  out=$(timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io-steps 0 $e 2>>gpurun_out/${tag}.err | tail -1)
  echo "$out" | python3 -c "
import json,sys
e=sys.argv[1]
try:
    d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms_per_call']
    print('%-40s %8.1f Mpts/s  calls %d x %.0f frames  stages %s  mism %s' % (e or '(default)', d['value'] or -1, d['config']['batch_calls'], d['config']['frames_per_call'], ' '.join('%.0f' % v for v in s.values()), (d.get('labels_checked') or {}).get('mismatches')))
except Exception as ex: print(e, 'failed', ex)
" "$e" | tee -a gpurun_out/${tag}.log
done
