import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import conftest, numpy as np
P = conftest.pkg(); ctx = P.Context(0)
def run(name, pts, prm, reps=2):
    for r in range(reps):
        t0 = time.perf_counter(); lab = ctx.segment(pts, prm); dt = time.perf_counter() - t0
    res = ctx.result
    tl = ctx.tile_list_lengths()
    print("   tiles %d, overflowed the LDS tables %d, one-ring list length median %d max %d" % (len(tl), int((tl == 0xFFFFFFFF).sum()), int(np.median(tl[tl != 0xFFFFFFFF])) if (tl != 0xFFFFFFFF).any() else -1,
                                                                                              int(tl[tl != 0xFFFFFFFF].max()) if (tl != 0xFFFFFFFF).any() else -1))
    print(name, "n", len(pts), "V", res.n_voxels, "S", res.n_supervoxels, "E", res.n_edges, "merges", res.n_merges, "regions", res.n_regions,
          "sweeps", res.sweeps, "ms %.1f" % (dt * 1e3), "stages", [round(x, 2) for x in res.ms_stage[:7]], flush=True)
run("cfg3 nyu 640x480", P.synth_frame(0, 77, 640, 480, 50), P.launch_params())
run("cfg2 1M", P.synth_frame(0, 1, 1000, 1000, 30), P.launch_params(voxel_res=0.008, seed_res=0.08))
run("cfg4 20M scene", P.synth_frame(1, 9, 5000, 4000, 0), P.launch_params(voxel_res=0.02, seed_res=0.2, use_transform=0))
