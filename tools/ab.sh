#!/bin/bash
# (GPU box) every same-box A/B of the rounds so far as ONE script: each run prints one line.  Replaces the single-use ab_*.sh / r3_*.sh / r4_*.sh.
#
#   tools/ab.sh env   "ENV=a ENV2=b" "-" ...     bench.py at the driver's flags under each environment ("-" = none): development switches, what-ifs
#   tools/ab.sh flags "--groups 8 --batch 144" ...   ... with extra bench.py flags per run (calls in flight, call size, ramp)
#   tools/ab.sh lib   <other.so> [rounds]        ... with the current library and another build of it, alternating (F3DS_LIB; path relative to the package dir)
#   tools/ab.sh iso   <kernel regex> <lib.so> ...   per-launch time of the matching kernels, ONE call of 192 frames at a time (rocprofv3 kernel trace), per build
#   tools/ab.sh one   "ENV=a" "-" ...            one call of 192 frames at a time (no other call in flight): per-stage device time of a call
#   tools/ab.sh lone  <lib.so> ...               merge stage of a lone 1M-point frame (8 waves; 4 waves, arrays in global memory) and of the 20M scene, per build, x3
#   tools/ab.sh hostio                           the host-in / host-out pass with and without the per-device copy stream
# Output also goes to gpurun_out/ab_<experiment>.log.  Lines: value [Mpoints/s], ms per step, merge launch ms, label mismatches, lone-frame latency, stage ms per call.
R=$PWD/fast-3d-pointcloud-segmentation_amd
export F3DS_DEV=1      # A/B runs use the development switches (csrc/f3ds_dev.h): their bench lines carry the rate as what_if_value, never as value
mkdir -p gpurun_out
DRV="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-io-steps 0"
line() { python3 -c "
import json,sys
tag=sys.argv[1]
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('%-48s %8.1f Mpts/s  %.2f ms/step  merge launch %.1f ms  mism %s  lone %.1f ms  calls %d x %.0f  stages %s' % (tag, d['value'] or d.get('what_if_value') or -1, d['ms_per_step'], r['launch_ms'],
          len((d.get('labels_checked') or {}).get('mismatches') or []), d['single_frame_latency_ms'], d['config']['batch_calls'], d['config']['frames_per_call'], ' '.join('%.1f' % v for v in r['stage_ms_per_call'].values())))
except Exception as ex: print(tag, 'failed', ex)
" "$1"; }
what=$1; shift
case "$what" in
env)   for e in "$@"; do [ "$e" = "-" ] && e=""; env $e timeout 500 python3 bench.py $DRV 2>>gpurun_out/ab.err | tail -1 | line "${e:-(default)}" | tee -a gpurun_out/ab_env.log; done ;;
flags) for e in "$@"; do [ "$e" = "-" ] && e=""; timeout 500 python3 bench.py $DRV --skip-latency $e 2>>gpurun_out/ab.err | tail -1 | line "${e:-(default)}" | tee -a gpurun_out/ab_flags.log; done ;;
lib)   other=$1; rounds=${2:-2}
       for r in $(seq $rounds); do for l in libf3ds.so $other; do F3DS_LIB=$R/$l timeout 500 python3 bench.py $DRV 2>>gpurun_out/ab.err | tail -1 | line "$l" | tee -a gpurun_out/ab_lib.log; done; done ;;
one)   for e in "$@"; do [ "$e" = "-" ] && e=""; env $e timeout 500 python3 bench.py --groups 1 --batch 192 --steps 9 --warmup 3 --no-cpu-baseline --host-io-steps 0 2>>gpurun_out/ab.err | tail -1 | line "${e:-(default)}" | tee -a gpurun_out/ab_one.log; done ;;
iso)   re=$1; shift; export TMPDIR=/tmp
       for l in "$@"; do
         rm -rf /tmp/iso_$$; (cd /tmp && F3DS_LIB=$R/$l timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/iso_$$ -- python3 $R/../bench.py --groups 1 --batch 192 --steps 6 --warmup 3 --host-io-steps 0 --no-cpu-baseline --skip-latency > /tmp/iso_$$.log 2>&1)
         python3 - "$re" "$l" /tmp/iso_$$ /tmp/iso_$$.log <<'PY' | tee -a gpurun_out/ab_iso.log
import csv,glob,sys,re,json
f=glob.glob(sys.argv[3]+"/**/*kernel_stats.csv",recursive=True)
tot=0; out=[]
for r in (csv.DictReader(open(f[0])) if f else []):
    tot+=float(r["TotalDurationNs"])
    m=re.search(r"d_[A-Za-z_0-9]+(<[^>]*>)?", r["Name"])
    if m and re.search(sys.argv[1], m.group(0)): out.append("%s %.3f (x%s)"%(m.group(0)[2:], float(r["AverageNs"])/1e6, r["Calls"]))
mm="?"
try: mm=len(json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])["labels_checked"]["mismatches"])
except Exception: pass
print("%-18s all kernels %.1f ms  mism %s | avg ms per launch: %s"%(sys.argv[2], tot/1e6, mm, "  ".join(sorted(out))))
PY
       done ;;
lone)  for rep in 1 2 3; do for l in "$@"; do
         echo -n "$l  8w: "; F3DS_LIB=$R/$l python3 tools/lone_frame.py 4 2>&1 | tail -2 | awk '{print $(NF-1)}' | tr '\n' ' '
         echo -n " 4w-global: "; F3DS_MERGE_NW=4 F3DS_MERGE_KEYS=global F3DS_LIB=$R/$l python3 tools/lone_frame.py 4 2>&1 | tail -2 | awk '{print $(NF-1)}' | tr '\n' ' '
         echo -n " config4 merge ms: "; F3DS_LIB=$R/$l python3 tools/config4_frame.py 3 2>&1 | grep "^scene" | tail -2 | sed 's/.*labels): //' | awk '{print $6}' | tr '\n' ' '; echo
       done; done | tee -a gpurun_out/ab_lone.log ;;
hostio) for r in 1 2; do for cs in 1 0; do F3DS_COPY_STREAM=$cs timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --skip-latency 2>>gpurun_out/ab.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('F3DS_COPY_STREAM=$cs', d['value'], d['value_survey_8d'], d['labels_checked']['mismatches'])" | tee -a gpurun_out/ab_hostio.log; done; done ;;
*) sed -n 2,12p "$0"; exit 2 ;;
esac
