#!/bin/bash
# host-in/host-out rate (bench.py value_host_io) with and without the per-device copy stream, alternating on one box
run() { echo -n "COPY_STREAM=$1 DIRECT=$2 : "; F3DS_COPY_STREAM=$1 F3DS_DIRECT_LABELS=$2 timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --skip-latency 2>>gpurun_out/ab.err | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['value_survey_8d'], d['labels_checked']['mismatches'], d.get('value_host_io'))
except Exception as e: print('failed', e)
"; }
for r in 1 2; do run 1 0; run 0 0; done
