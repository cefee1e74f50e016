#!/bin/bash
# On the GPU box (through gpurun): everything profiles/<tag>_* is built from, into gpurun_out/<tag>_*.
#   tools/refresh_profiles.sh <tag>
# Then here: cp the summaries named at the end of this script into profiles/.
tag=${1:-r6}
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
# 1. the bench line as the driver runs it, and with the default flags
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_flags.json 2> gpurun_out/${tag}_bench.err
timeout 600 python3 bench.py > gpurun_out/${tag}_bench.json 2>> gpurun_out/${tag}_bench.err
# 2. per-kernel times of the same command (six calls in flight: durations include sharing the chip) + what the GPU does over time
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_stats -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --host-io-steps 0 --no-cpu-baseline > $R/gpurun_out/${tag}_stats.log 2>&1
cd $R; cp $(find /tmp/${tag}_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
python3 tools/timeline.py /tmp/${tag}_stats calls=12 > gpurun_out/${tag}_timeline.txt 2>&1
grep -v "^[EWI]2026" gpurun_out/${tag}_stats.log | grep "^{" | tail -1 > gpurun_out/${tag}_bench_profiled_run.json      # the bench line of the PROFILED run: its launch_ms is compared with the trace of the same run
# 3. one call of 192 frames at a time: per-kernel cost without contention, and the sweep kernels launch by launch
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_iso -- python3 $R/bench.py --groups 1 --batch 192 --steps 9 --warmup 3 --host-io-steps 0 --no-cpu-baseline --skip-latency > $R/gpurun_out/${tag}_iso.log 2>&1
cd $R; cp $(find /tmp/${tag}_iso -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats_isolated.csv
python3 tools/sweep_trace.py /tmp/${tag}_iso 192 > gpurun_out/${tag}_sweep_trace.txt 2>&1
# 4. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (one call of 96 frames at a time, equal calls)
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && F3DS_BENCH_RAMP=0 timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/${tag}_pmc_$c -- python3 $R/bench.py --groups 1 --batch 96 --steps 3 --warmup 0 --host-io-steps 0 --no-cpu-baseline --skip-latency > $R/gpurun_out/${tag}_pmc_$c.log 2>&1
done
cd $R
# the profiled run processes: 2 set-up passes of 96 + 3 steps of 64 (no latency frames: --skip-latency)
python3 tools/pmc_summary.py /tmp/${tag}_pmc_FETCH_SIZE /tmp/${tag}_pmc_WRITE_SIZE 96 gpurun_out/${tag}_pmc_hbm_traffic.json 384 > gpurun_out/${tag}_pmc_summary.txt 2>&1
tail -20 gpurun_out/${tag}_pmc_summary.txt
# 5. BASELINE config 4 (the 20M-point scene): stage times, per-kernel stats, merge-loop phase probes (PROF build), HBM traffic
python3 tools/config4_frame.py 4 > gpurun_out/${tag}_config4_stages.txt 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_c4 -- python3 $R/tools/config4_frame.py 4 > $R/gpurun_out/${tag}_config4.log 2>&1
cd $R; cp $(find /tmp/${tag}_c4 -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_config4_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/${tag}_c4pmc_$c -- python3 $R/tools/config4_frame.py 2 > $R/gpurun_out/${tag}_c4pmc_$c.log 2>&1
done
cd $R
python3 tools/pmc_summary.py /tmp/${tag}_c4pmc_FETCH_SIZE /tmp/${tag}_c4pmc_WRITE_SIZE 1 gpurun_out/${tag}_config4_pmc_hbm_traffic.json 2 > gpurun_out/${tag}_config4_pmc_summary.txt 2>&1
tail -12 gpurun_out/${tag}_config4_pmc_summary.txt
if [ -f fast-3d-pointcloud-segmentation_amd/libf3ds_prof.so ]; then
  F3DS_DEV=1 F3DS_LIB=$R/fast-3d-pointcloud-segmentation_amd/libf3ds_prof.so python3 tools/lone_frame.py 3 > gpurun_out/${tag}_merge_prof_raw.txt 2>&1
  F3DS_DEV=1 F3DS_LIB=$R/fast-3d-pointcloud-segmentation_amd/libf3ds_prof.so python3 tools/config4_frame.py 2 > gpurun_out/${tag}_config4_merge_prof_raw.txt 2>&1
fi
# 6. SQ counters of the wide kernels, one call of 192 frames at a time (rocprofv3 serialises the dispatches it counts): issue / wait split, then LDS
cd /tmp
timeout 700 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/${tag}_sqA -- python3 $R/bench.py --groups 1 --batch 192 --steps 2 --warmup 1 --host-io-steps 0 --no-cpu-baseline --skip-latency > $R/gpurun_out/${tag}_sqA.log 2>&1
timeout 700 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/${tag}_sqB -- python3 $R/bench.py --groups 1 --batch 192 --steps 2 --warmup 1 --host-io-steps 0 --no-cpu-baseline --skip-latency > $R/gpurun_out/${tag}_sqB.log 2>&1
cd $R
K="d_normals_t d_sweep_claim d_centroid d_sweep_R d_neighbors d_tile_keys d_tile_place d_voxel_list_accum d_merge_il_t"
{ echo "# pass A: issue / wait split (per launch of 192 frames, largest-grid launches)"; python3 tools/pmc_sq.py /tmp/${tag}_sqA $K; echo "# pass B: LDS"; python3 tools/pmc_sq.py /tmp/${tag}_sqB $K; } > gpurun_out/${tag}_sq_counters.txt 2>&1
# 7. memory-pipeline counters of the wide kernels (TA / TCP / TCC), one call of 192 frames at a time
tools/pmc_mem.sh d_sweep_claim d_sweep_R_first d_sweep_R d_centroid d_normals d_neighbors d_sv_fill d_voxel_list d_tile_keys > gpurun_out/${tag}_pmc_mem_run.log 2>&1
cp gpurun_out/pmc_mem.txt gpurun_out/${tag}_mem_pipeline_counters.txt
echo "copy to profiles/: ${tag}_bench.json ${tag}_bench_driver_flags.json ${tag}_kernel_stats.csv ${tag}_kernel_stats_isolated.csv ${tag}_timeline.txt ${tag}_sweep_trace.txt ${tag}_pmc_hbm_traffic.json ${tag}_sq_counters.txt ${tag}_config4_*"
