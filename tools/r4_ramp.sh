#!/bin/bash
run() { echo -n "$* : "; env $ENVV timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io-steps 0 --skip-latency "$@" 2>>gpurun_out/ramp.err | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['batch_calls'], d['labels_checked']['mismatches'])
except Exception as e: print('failed', e)
"; }
for r in 1 2; do
ENVV="A=1" run --batch 192
ENVV="F3DS_BENCH_RAMP=0" run --batch 192
ENVV="A=1" run --batch 160
ENVV="A=1" run --batch 224
ENVV="A=1" run --batch 128
done
