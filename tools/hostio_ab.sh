#!/bin/bash
export F3DS_DEV=1
for r in 1 2 3; do for cs in 1 2; do F3DS_COPY_STREAM=$cs timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --skip-latency 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['value_host_io']; print('F3DS_COPY_STREAM=$cs', d['what_if_value'], h['value'], h['link_bound']['both_ways_at_once_GBps_each'], d['labels_checked']['mismatches'])"; done; done
