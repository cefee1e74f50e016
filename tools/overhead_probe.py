"""Per-frame fixed cost (host API + dispatch) of the path: tiny frames, so GPU work is negligible.
usage: python tools/overhead_probe.py BATCH GROUPS [steps]"""
import importlib, os, sys, threading, time, queue
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("fast-3d-pointcloud-segmentation_amd")
B = int(sys.argv[1]); G = int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 512
prm = P.launch_params(voxel_res=0.02, seed_res=0.2)
fr = torch.from_numpy(P.synth_frame(0, 7, 160, 120, 30)).cuda(); n = 160 * 120
ctxs = [[P.Context(0) for _ in range(B)] for _ in range(G)]
outs = [[torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(B)] for _ in range(G)]
def run(count):
    q = queue.Queue()
    for s in range(0, count, B): q.put(min(B, count - s))
    def w(g):
        while True:
            try: k = q.get_nowait()
            except queue.Empty: return
            P.segment_batch(ctxs[g][:k], [fr.data_ptr()] * k, prm, labels_out=[outs[g][i].data_ptr() for i in range(k)], n=[n] * k, on_device=True)
    th = [threading.Thread(target=w, args=(g,)) for g in range(G)]
    [t.start() for t in th]; [t.join() for t in th]
run(B * G); torch.cuda.synchronize()
t = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("tiny frames: batch=%d groups=%d: %.3f ms/frame" % (B, G, dt / steps * 1e3))
