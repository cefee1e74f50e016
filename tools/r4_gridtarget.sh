run() { echo -n "$1 : "; env $1 timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io-steps 0 --skip-latency 2>>gpurun_out/gt.err | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['labels_checked']['mismatches'], [round(v,1) for v in d['roofline']['stage_ms_per_call'].values()])
except Exception as e: print('failed', e)
"; }
for r in 1 2; do run F3DS_GRID_TARGET=3072; run F3DS_GRID_TARGET=2048; run F3DS_GRID_TARGET=4608; run F3DS_GRID_TARGET=6144; done
