#!/bin/bash
# scheduling what-ifs on one box with the list-walking merge kernel (52 KB of LDS per merge workgroup): calls in flight, call sizes, merge width
run() { echo -n "$* : "; env "${ENVV[@]}" timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io-steps 0 "$@" 2>>gpurun_out/sched.err | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], list(d['roofline']['merge_layouts_last_call'].items()), d['labels_checked']['mismatches'], [round(v,1) for v in d['roofline']['stage_ms_per_call'].values()])
except Exception as e: print('failed', e)
"; }
ENVV=(A=1); run --groups 6
ENVV=(A=1); run --groups 8
ENVV=(A=1); run --groups 8 --batch 144
ENVV=(A=1); run --groups 7
ENVV=(F3DS_MERGE_NW=8 F3DS_MERGE_KEYS=global); run --groups 6
ENVV=(A=1); run --groups 6
ENVV=(A=1); run --groups 6 --batch 256
ENVV=(A=1); run --groups 5 --batch 256
ENVV=(A=1); run --groups 10 --batch 128
