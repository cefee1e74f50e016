#!/bin/bash
# (GPU box) memory-pipeline counters of the wide kernels, one call of 192 frames at a time: TA / TCP / TCC busy and stall cycles per launch.
# Two counters per block and pass (more: "exceeds the capabilities of the hardware", and the profiled program hangs); stops at the first pass that fails.
# usage: tools/pmc_mem.sh [kernel filter ...]     -> gpurun_out/pmc_mem.txt
R=$PWD; export TMPDIR=/tmp
: > gpurun_out/pmc_mem.txt
run() { tag=$1; shift; rm -rf /tmp/pm_$tag; cd /tmp; timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pm_$tag -- python3 $R/bench.py --groups 1 --batch 192 --steps 2 --warmup 1 --host-io-steps 0 --no-cpu-baseline --skip-latency > $R/gpurun_out/pm_$tag.log 2>&1; rc=$?; cd $R
  if [ $rc -ne 0 ]; then echo "pass $tag ($*) failed rc=$rc"; grep -m1 "error code" gpurun_out/pm_$tag.log; exit 1; fi
  { echo "# pass $tag"; python3 tools/pmc_sq.py /tmp/pm_$tag $FILTER 2>&1 | grep -v "^   per wave\|^   LDS"; } >> gpurun_out/pmc_mem.txt; }
FILTER="$*"
run A GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum
run B GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run C GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum
run D GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run E GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run F GRBM_GUI_ACTIVE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum
