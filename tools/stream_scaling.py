"""Throughput of independent f3ds_segment calls on S streams (no batching): probes how many
dispatches from different HIP streams really overlap.  usage: GPU_MAX_HW_QUEUES=q python tools/stream_scaling.py S [steps]"""
import importlib, os, sys, threading, time, queue
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
S = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
prm = P.launch_params()
frames = [torch.from_numpy(P.synth_frame(0, 1000 + i, 1000, 1000, 30)).cuda() for i in range(4)]
ctxs = [P.Context(0) for _ in range(S)]
outs = [torch.empty(1000000, dtype=torch.int32, device="cuda") for _ in range(S)]
def run(count):
    q = queue.Queue()
    for s in range(count): q.put(s)
    def w(i):
        while True:
            try: s = q.get_nowait()
            except queue.Empty: return
            ctxs[i].segment(frames[s % 4].data_ptr(), prm, labels_out=outs[i].data_ptr(), n=1000000, on_device=True)
    th = [threading.Thread(target=w, args=(i,)) for i in range(S)]
    [t.start() for t in th]; [t.join() for t in th]
run(S * 2); torch.cuda.synchronize()
t = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("HWQ=%s streams=%d: %.1f Mpts/s (%.2f ms/frame)" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), S, steps / dt, dt / steps * 1e3))
