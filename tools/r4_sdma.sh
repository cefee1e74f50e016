#!/bin/bash
run() { echo -n "$* : "; env "$@" timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --skip-latency 2>>gpurun_out/ab.err | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['value_survey_8d'], d['labels_checked']['mismatches'])
except Exception as e: print('failed', e)
"; }
run HSA_ENABLE_SDMA=1
run HSA_ENABLE_SDMA=0
run HSA_ENABLE_SDMA=0 F3DS_COPY_STREAM=0
run HSA_ENABLE_SDMA=1
