#!/bin/bash
# (GPU box) bench at the driver's flags for several (calls in flight, frames per call) settings
for gb in "6 192" "7 192" "8 192" "6 128" "8 128" "5 192" "6 192"; do set -- $gb
  echo -n "groups $1 batch $2 : "; timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --groups $1 --batch $2 --no-cpu-baseline --host-io-steps 0 --skip-latency 2>>gpurun_out/ab.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [round(v,1) for v in d['roofline']['stage_ms_per_call'].values()])"
done
