#!/bin/bash
# (GPU box) per-launch time of one kernel with one call of 192 frames at a time, for the current library and another build: tools/iso_kernel.sh <kernel regex> <other .so>
R=$PWD/fast-3d-pointcloud-segmentation_amd; export TMPDIR=/tmp
for lib in libf3ds.so $2 libf3ds.so $2; do
  rm -rf /tmp/iso_$$; cd /tmp
  F3DS_LIB=$R/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/iso_$$ -- python3 $R/../bench.py --groups 1 --batch 192 --steps 6 --warmup 3 --host-io-steps 0 --no-cpu-baseline --skip-latency > /tmp/iso_$$.log 2>&1
  cd $R/..
  python3 - "$1" "$lib" /tmp/iso_$$ <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[3]+"/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    tot+=float(r["TotalDurationNs"])
    import re
    m=re.search(r"d_[A-Za-z_0-9]+(<[^>]*>)?", r["Name"])
    if m and re.search(sys.argv[1], m.group(0)): print(sys.argv[2], "%-22s"%m.group(0), "calls", r["Calls"], "total ms %.2f"%(float(r["TotalDurationNs"])/1e6), "avg ms %.3f"%(float(r["AverageNs"])/1e6))
print(sys.argv[2], "all kernels ms", round(tot/1e6,1))
PY
done
