#!/bin/bash
# (GPU box) the bench at the driver's flags under the development switches whose defaults were chosen earlier in the round: are they still the best with the final kernels?
run() { echo -n "${1:-default} : "; env $1 timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io-steps 0 --skip-latency 2>>gpurun_out/ab.err | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['labels_checked']['mismatches'], [round(v,1) for v in d['roofline']['stage_ms_per_call'].values()])
except Exception as e: print('failed', e)
"; }
for r in 1 2; do run ""; run F3DS_MERGE_SHARED_RES=2; run F3DS_NORMALS_THREADS=384; run F3DS_MERGE_SPEC=0; done
