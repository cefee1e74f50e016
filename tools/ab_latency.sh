#!/bin/bash
# usage: tools/ab_latency.sh <lib.so> ...   -- single-frame latency and merge-stage time of each library build, same box
for rep in 1 2; do
for lib in "$@"; do
  F3DS_LIB=$PWD/$lib python3 bench.py --batch 1 --groups 1 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'latency', d['single_stream_latency_ms'], 'merge', d['roofline']['stage_ms']['merge'])"
done; done
