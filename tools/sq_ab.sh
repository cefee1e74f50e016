#!/bin/bash
# (GPU box) SQ counters (issue / wait split, instruction mix) and the sweep kernels launch by launch, per build, one call of 192 frames at a time.
#   tools/sq_ab.sh "<kernel names for pmc_sq.py>" <lib.so> [<lib.so> ...]       (libraries relative to the package directory)
R=$PWD; export F3DS_DEV=1 TMPDIR=/tmp
K=$1; shift
for l in "$@"; do
  echo "=== $l"
  rm -rf /tmp/sqab /tmp/sqtr
  (cd /tmp && F3DS_LIB=$R/fast-3d-pointcloud-segmentation_amd/$l timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/sqab -- python3 $R/bench.py --groups 1 --batch 192 --steps 2 --warmup 1 --host-io-steps 0 --no-cpu-baseline --skip-latency > /tmp/sqab.log 2>&1)
  python3 tools/pmc_sq.py /tmp/sqab $K
  (cd /tmp && F3DS_LIB=$R/fast-3d-pointcloud-segmentation_amd/$l timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/sqtr -- python3 $R/bench.py --groups 1 --batch 192 --steps 3 --warmup 1 --host-io-steps 0 --no-cpu-baseline --skip-latency > /tmp/sqtr.log 2>&1)
  python3 tools/sweep_trace.py /tmp/sqtr 192
done
