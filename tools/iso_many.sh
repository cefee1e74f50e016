#!/bin/bash
# (GPU box) per-launch time of the kernels matching a regex, one call of 192 frames at a time, for several builds: tools/iso_many.sh <kernel regex> <lib> [<lib> ...]
R=$PWD/fast-3d-pointcloud-segmentation_amd; export TMPDIR=/tmp
re=$1; shift
for lib in "$@"; do
  rm -rf /tmp/iso_$$; cd /tmp
  F3DS_LIB=$R/$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/iso_$$ -- python3 $R/../bench.py --groups 1 --batch 192 --steps 6 --warmup 3 --host-io-steps 0 --no-cpu-baseline --skip-latency > /tmp/iso_$$.log 2>&1
  cd $R/..
  python3 - "$re" "$lib" /tmp/iso_$$ /tmp/iso_$$.log <<'PY'
import csv,glob,sys,re,json
f=glob.glob(sys.argv[3]+"/**/*kernel_stats.csv",recursive=True)[0]
tot=0; out=[]
for r in csv.DictReader(open(f)):
    tot+=float(r["TotalDurationNs"])
    m=re.search(r"d_[A-Za-z_0-9]+(<[^>]*>)?", r["Name"])
    if m and re.search(sys.argv[1], m.group(0)): out.append("%s %.3f"%(m.group(0)[2:], float(r["AverageNs"])/1e6))
mm="?"
try: mm=len(json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])["labels_checked"]["mismatches"])
except Exception as e: pass
print("%-16s all %.1f ms mism %s | %s"%(sys.argv[2], tot/1e6, mm, "  ".join(sorted(out))))
PY
done
