#!/bin/bash
# On the GPU box (through gpurun): everything profiles/ is built from, into gpurun_out/<tag>_*.
#   tools/refresh_profiles.sh <tag>
# Then here: cp gpurun_out/<tag>_bench.json profiles/r1_bench.json, the *_kernel_stats.csv of <tag>_stats and
# <tag>_stats_b32, and tools/pmc_summary.py gpurun_out/<tag>_pmc_fetch gpurun_out/<tag>_pmc_write 96 profiles/r1_pmc_hbm_traffic.json
tag=${1:-r1}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 1536 --warmup 768 --no-cpu-baseline > gpurun_out/${tag}_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_b32 -- python3 bench.py --batch 32 --groups 1 --steps 128 --warmup 32 --no-cpu-baseline > gpurun_out/${tag}_stats_b32.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python3 bench.py --batch 96 --groups 1 --steps 96 --warmup 96 --no-cpu-baseline > gpurun_out/${tag}_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_write -- python3 bench.py --batch 96 --groups 1 --steps 96 --warmup 96 --no-cpu-baseline > gpurun_out/${tag}_pmc_write.log 2>&1
# the raw traces are large: keep the summaries and the counter tables only
find gpurun_out/${tag}_stats gpurun_out/${tag}_stats_b32 -name "*kernel_trace.csv" -delete
find gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write -name "*kernel_trace.csv" -delete
echo done
