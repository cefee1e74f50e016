#!/bin/bash
# bench A/B of two builds of the library on one box, alternating: tools/ab_lib.sh <other .so (relative to the package dir)> [rounds]
R=$PWD/fast-3d-pointcloud-segmentation_amd
other=$1; rounds=${2:-2}
run() { echo -n "$1 : "; F3DS_LIB=$R/$1 timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io-steps 0 2>>gpurun_out/ab.err | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['labels_checked']['mismatches'], d['single_frame_latency_ms'], [round(v,1) for v in d['roofline']['stage_ms_per_call'].values()])
except Exception as e: print('failed', e)
"; }
for r in $(seq $rounds); do run libf3ds.so; run $other; done
