F3DS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --host-io-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 5 --host-io-steps 0 2>/dev/null | tail -1 | cut -c1-200
