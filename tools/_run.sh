timeout 200 python tools/time_configs.py 2>&1 | tee gpurun_out/r2v_time_configs.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2v_gputest.log 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r2v_gputest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
