timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2l_gputest.log 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r2l_gputest.log | tail -5
F3DS_LIB=$PWD/fast-3d-pointcloud-segmentation_amd/libf3ds_prof.so timeout 200 python tools/merge_variants.py 1 2>&1 | grep "d_merge_cw" | sort -u -k1,1 | tee gpurun_out/r2l_prof1.log
timeout 100 python tools/merge_variants.py 1 2>&1 | tee gpurun_out/r2l_variants1.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2l_bench_driver.json 2>gpurun_out/r2l_bench_driver.err; cut -c1-400 gpurun_out/r2l_bench_driver.json
