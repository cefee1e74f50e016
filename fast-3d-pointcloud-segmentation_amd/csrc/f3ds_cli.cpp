// supervoxel_clustering -- drop-in command line for the segmentation path of the reference
// (/root/reference/src/supervoxel_clustering.cpp:136-476): same flags, defaults, mutual-exclusion
// rule and exit codes; the VCCS + Clustering work runs in libf3ds (HIP, MI355X) through the C-ABI.
//
// Kept from the reference: -d/-p, -v -s -c -z -n, -t, --RGB --CVX --ML --AL --EQ, -r, -f, --NT, --V.
// Added (additive): -o <pcd> coloured voxel cloud (Clustering::get_colored_cloud), --labels <file>
// per-point uint32 region ids, --gpu <id>, --gpus <N> (label files only: the files are sharded over N GPUs, file i on GPU i mod N, one host thread per
// GPU, labels gathered on GPU 0 over RCCL: f3ds_multi_*), --dump <dir> (what visualize() draws, as PCD files), --refine <n> (refineSupervoxels, :369-375), --stream <depth> (label files only: the files go through the frame
// pipeline f3ds_stream_*, reading ahead of the GPU, no evaluation).  Without -t the threshold is chosen by the ground-truth sweep
// (all_thresh 0.8..1 step 0.005 + best_thresh, :428-437) and the <-f name>_*.csv score files are written
// (:471, manageAllPerformances); every file is scored against its `label` field (:462-463).
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <string>
#include <vector>

#include "../../include/f3ds.h"

namespace {
// pcl::console semantics: first exact match; value = atof/atoi of the next token
int find_argument(int argc, char** argv, const char* name) {
    for (int i = 1; i < argc; ++i)
        if (strcmp(argv[i], name) == 0) return i;
    return -1;
}
bool find_switch(int argc, char** argv, const char* name) { return find_argument(argc, argv, name) != -1; }
void parse(int argc, char** argv, const char* name, float& v) {
    int i = find_argument(argc, argv, name);
    if (i > 0 && i + 1 < argc) v = (float)atof(argv[i + 1]);
}
void parse(int argc, char** argv, const char* name, int& v) {
    int i = find_argument(argc, argv, name);
    if (i > 0 && i + 1 < argc) v = atoi(argv[i + 1]);
}
void parse(int argc, char** argv, const char* name, std::string& v) {
    int i = find_argument(argc, argv, name);
    if (i > 0 && i + 1 < argc) v = argv[i + 1];
}
// PCD v0.7 binary file of pcl::PointNormal-like records (x y z normal_x normal_y normal_z): makeSupervoxelNormalCloud's output
bool write_normal_pcd(const std::string& path, const std::vector<float>& xyz, const std::vector<float>& nrm, size_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z\nSIZE 4 4 4 4 4 4\nTYPE F F F F F F\nCOUNT 1 1 1 1 1 1\n"
               "WIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n", n, n);
    for (size_t i = 0; i < n; ++i) { fwrite(&xyz[3 * i], 4, 3, f); fwrite(&nrm[3 * i], 4, 3, f); }
    const bool ok = !ferror(f);
    fclose(f);
    return ok;
}
bool verbose = false;
#define DEBUG(...) do { if (verbose) fprintf(stderr, __VA_ARGS__); } while (0)
struct P16 { float x, y, z; uint32_t rgba; };
}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        printf("Syntax is: %s {-d <direcory-of-pcd-files> OR -p <pcd-file>} [arguments] \n\n\t"
               "SUPERVOXEL optional arguments: \n\t"
               " -v <voxel-resolution>          (default: 0.008) \n\t"
               " -s <seed-resolution>           (default: 0.08) \n\t"
               " -c <color-weight>              (default: 0.2) \n\t"
               " -z <spatial-weight>            (default: 0.4) \n\t"
               " -n <normal-weight>             (default: 1.0) \n\t\n\t"
               "SEGMENTATION optional arguments: \n\t"
               " -t <threshold>                 (default: auto)\n\t"
               " --RGB                          (RGB euclidean colour distance instead of L*A*B* CIEDE2000) \n\t"
               " --CVX                          (convexity criterion weighs the geometric distance) \n\t"
               " --ML [manual-lambda] *         (Manual Lambda merging; lambda=0.5 if no value) \n\t"
               " --AL                 *         (Adaptive Lambda merging) \n\t"
               " --EQ [bins-number]   *         (Equalization merging; default bins if no value) \n\t"
               "  * only one of these can be passed at a time \n\t\n\t"
               "OTHER optional arguments: \n\t"
               " -r <label-to-be-removed>       (drops points with this ground-truth label) \n\t"
               " -f <test-results-filename>     (accepted for compatibility) \n\t"
               " --NT                           (disables the single camera transform) \n\t"
               " --V                            (verbose) \n\t"
               " -o <out.pcd>                   (writes the coloured voxel cloud) \n\t"
               " --labels <file>                (writes per-point uint32 region ids) \n\t"
               " --gpu <id>                     (HIP device, default 0) \n\t"
               " --gpus <N>                     (with -t and --labels: files sharded over N GPUs, labels gathered on GPU 0 over RCCL) \n\t"
               " --dump <dir>                   (voxel centroids, supervoxel normals, adjacency graph, refined cloud as PCD: what the viewer shows) \n\t"
               " --stream <depth>               (with -t and --labels: files through the frame pipeline, <depth> in flight) \n\t"
               " --refine <iterations>          (refineSupervoxels as main() does with 3, :369-375; with -o also <out.pcd>.refined) \n\t"
               " --bench <frames>               (no input files: <frames> synthetic 1M-point RGB-D frames through the path, with --gpus N sharded and pipelined; prints Mpoints/s) \n",
               argv[0]);
        return 1;
    }
    verbose = find_switch(argc, argv, "--V");
    const bool disable_transform = find_switch(argc, argv, "--NT");
    std::string test_filename = "test";
    if (find_switch(argc, argv, "-f")) parse(argc, argv, "-f", test_filename);
    std::string path;
    std::vector<std::string> file_list;
    if (find_switch(argc, argv, "-d")) {
        parse(argc, argv, "-d", path);
        printf("Counting files in directory...\n");
        std::error_code ec;
        if (!std::filesystem::exists(path, ec) || !std::filesystem::is_directory(path, ec)) {
            fprintf(stderr, "Specified directory doesn't exists or can't be opened\n");
            return 1;
        }
        for (auto it = std::filesystem::recursive_directory_iterator(path, ec); !ec && it != std::filesystem::recursive_directory_iterator(); ++it)
            if (it->is_regular_file() && it->path().extension() == ".pcd") { file_list.push_back(it->path().string()); DEBUG("File found: %s\n", it->path().c_str()); }
        printf("Found %zu files\n", file_list.size());
    } else if (find_switch(argc, argv, "-p")) {
        parse(argc, argv, "-p", path);
        file_list.push_back(path);
    } else if (!find_switch(argc, argv, "--bench")) {
        fprintf(stderr, "No input file or directory specified\n");
        return 1;
    }
    const bool thresh_specified = find_switch(argc, argv, "-t");
    f3ds_params prm;
    f3ds_default_params(&prm);
    float thresh = 0;
    if (thresh_specified) { parse(argc, argv, "-t", thresh); DEBUG("Using threshold: %f\n", thresh); }
    else DEBUG("Using automatic threshold\n");
    if (find_switch(argc, argv, "-v")) parse(argc, argv, "-v", prm.voxel_res);
    if (find_switch(argc, argv, "-s")) parse(argc, argv, "-s", prm.seed_res);
    if (find_switch(argc, argv, "-c")) parse(argc, argv, "-c", prm.w_color);
    if (find_switch(argc, argv, "-z")) parse(argc, argv, "-z", prm.w_spatial);
    if (find_switch(argc, argv, "-n")) parse(argc, argv, "-n", prm.w_normal);
    const bool rgb = find_switch(argc, argv, "--RGB"), cvx = find_switch(argc, argv, "--CVX");
    bool ml = find_switch(argc, argv, "--ML"), al = find_switch(argc, argv, "--AL"), eq = find_switch(argc, argv, "--EQ");
    if (!(ml || al || eq)) { al = true; DEBUG("No merging criterion specified, Adaptive Lambda is going to be used\n"); }
    else if (!(ml ^ al ^ eq)) { fprintf(stderr, "Only one parameter between --ML --AL and --EQ can be specified at a time\n"); return 1; }
    float lambda = 0; if (ml) parse(argc, argv, "--ML", lambda);
    int bin_num = 0; if (eq) parse(argc, argv, "--EQ", bin_num);
    const bool remove_label = find_switch(argc, argv, "-r");
    int label_to_be_removed = 0; if (remove_label) parse(argc, argv, "-r", label_to_be_removed);
    std::string out_pcd, out_labels; int gpu = 0;
    if (find_switch(argc, argv, "-o")) parse(argc, argv, "-o", out_pcd);
    if (find_switch(argc, argv, "--labels")) parse(argc, argv, "--labels", out_labels);
    if (find_switch(argc, argv, "--gpu")) parse(argc, argv, "--gpu", gpu);
    prm.use_transform = !disable_transform;
    prm.color_metric = rgb ? F3DS_RGB_EUCL : F3DS_LAB_CIEDE00;
    prm.geom_metric = cvx ? F3DS_CONVEX_NORMALS_DIFF : F3DS_NORMALS_DIFF;
    prm.merging = ml ? F3DS_MANUAL_LAMBDA : (eq ? F3DS_EQUALIZATION : F3DS_ADAPTIVE_LAMBDA);
    prm.lambda = lambda; prm.bins = bin_num; prm.threshold = thresh; prm.fold_negative_z = 1;
    int refine_itr = 0;
    if (find_switch(argc, argv, "--refine")) parse(argc, argv, "--refine", refine_itr);
    std::string dump_dir;
    if (find_switch(argc, argv, "--dump")) parse(argc, argv, "--dump", dump_dir);
    if (!dump_dir.empty() && refine_itr <= 0) refine_itr = 3;      // main() refines with 3 iterations for the normals the viewer shows (:371)
    int stream_depth = 0;
    if (find_switch(argc, argv, "--stream")) parse(argc, argv, "--stream", stream_depth);
    if (stream_depth > 0) {
        // Streaming mode: what a ROS node around this path would do (README.md:72) -- frames in, per-point labels out, in
        // order; file k+1 is read (into the pinned slot) while the GPU works on file k.  No ground-truth sweep or scores here.
        if (!thresh_specified || out_labels.empty() || remove_label) { fprintf(stderr, "--stream needs -t <threshold> and --labels <file>, and does not take -r\n"); return 1; }
        f3ds_stream* fs = nullptr;
        int rc = f3ds_stream_create(gpu, stream_depth, 0, &fs);
        if (rc) { fprintf(stderr, "f3ds_stream_create: %s %s\n", f3ds_strerror(rc), f3ds_last_hip_error()); return 1; }
        int failed = 0;
        auto take = [&](int wait) -> bool {                    // one finished frame -> its label file
            size_t n = 0; uint64_t tag = 0; f3ds_result res;
            static std::vector<uint32_t> labels;
            uint32_t probe;
            int r = f3ds_stream_next(fs, &probe, 0, &n, &tag, &res, wait);
            if (r == F3DS_ERR_EMPTY || r == F3DS_ERR_BUSY) return false;
            if (r == F3DS_ERR_CAPACITY) { labels.resize(n); r = f3ds_stream_next(fs, labels.data(), n, &n, &tag, &res, 1); }
            const std::string& file = file_list[tag];
            if (r) { fprintf(stderr, "%s: %s\n", file.c_str(), f3ds_strerror(r)); failed = 1; return true; }
            std::string suffix = file_list.size() > 1 ? "." + std::filesystem::path(file).stem().string() : "";
            FILE* f = fopen((out_labels + suffix).c_str(), "wb");
            if (!f || fwrite(labels.data(), 4, n, f) != n) { fprintf(stderr, "writing %s failed\n", (out_labels + suffix).c_str()); failed = 1; }
            if (f) fclose(f);
            printf("%s: %llu points, %u voxels, %u supervoxels, %u merges -> %u regions\n", file.c_str(), (unsigned long long)res.n_points, res.n_voxels,
                   res.n_supervoxels, res.n_merges, res.n_regions);
            return true;
        };
        for (size_t k = 0; k < file_list.size(); ++k) {
            size_t n = 0; void* buf = nullptr;
            if (f3ds_pcd_read(file_list[k].c_str(), nullptr, nullptr, 0, &n, nullptr, nullptr) != F3DS_OK) n = 0;
            while ((rc = f3ds_stream_buffer(fs, n, &buf)) == F3DS_ERR_BUSY) take(1);
            if (rc) { fprintf(stderr, "f3ds_stream_buffer: %s\n", f3ds_strerror(rc)); failed = 1; break; }
            if (n && f3ds_pcd_read(file_list[k].c_str(), buf, nullptr, n, &n, nullptr, nullptr) != F3DS_OK) n = 0;      // straight into pinned memory
            if ((rc = f3ds_stream_submit(fs, buf, n, &prm, k))) { fprintf(stderr, "f3ds_stream_submit: %s\n", f3ds_strerror(rc)); failed = 1; break; }
            while (take(0)) {}
        }
        while (f3ds_stream_pending(fs) > 0) take(1);
        f3ds_stream_destroy(fs);
        return failed;
    }
    int gpus = 0;
    if (find_switch(argc, argv, "--gpus")) parse(argc, argv, "--gpus", gpus);
    if (find_switch(argc, argv, "--bench")) {
        // Throughput of the path on synthetic frames (BASELINE.json config 2 / 5: 1000 x 1000 pinhole RGB-D frames, 3 % invalid depth,
        // seeds 1000 ...), host buffers in, per-point labels in host buffers out.  With --gpus N: frame i -> GPU i mod N through the
        // pipelined multi-GPU driver (chunks of 8 frames per GPU; chunk k+1 computes while chunk k's labels are gathered over RCCL
        // and copied out).  The reference has no counterpart; bench.py is the harness the measured numbers of this repository come from.
        int frames = 64;
        parse(argc, argv, "--bench", frames);
        if (frames <= 0 || !thresh_specified) { fprintf(stderr, "--bench <frames> needs a positive frame count and -t <threshold>\n"); return 1; }
        const uint32_t W = 1000, H = 1000; const size_t npts = (size_t)W * H;
        const int distinct = std::min(frames, 16);             // the frames cycle through 16 distinct seeds
        std::vector<std::vector<P16>> pts((size_t)distinct, std::vector<P16>(npts));
        for (int i = 0; i < distinct; ++i) if (f3ds_synth_frame(0, 1000u + (uint64_t)i, W, H, 30, pts[(size_t)i].data())) { fprintf(stderr, "f3ds_synth_frame failed\n"); return 1; }
        const int G = gpus > 0 ? gpus : 1, per_device = 8, chunk = G * per_device;
        f3ds_multi* mg = nullptr;
        int rc = f3ds_multi_create(nullptr, G, per_device, &mg);
        if (!rc) rc = f3ds_multi_reserve(mg, npts);
        if (rc) { fprintf(stderr, "f3ds_multi_create(%d GPUs): %s %s %s\n", G, f3ds_strerror(rc), f3ds_last_hip_error(), f3ds_multi_last_error()); return 1; }
        std::vector<std::vector<uint32_t>> labels[2];
        for (auto& l : labels) l.assign((size_t)chunk, std::vector<uint32_t>(npts));
        std::vector<f3ds_result> res[2]; res[0].resize((size_t)chunk); res[1].resize((size_t)chunk);      // written by the driver's threads until a batch is collected: they outlive run()
        auto run = [&](int total, double* seconds, f3ds_result* last) -> int {
            const auto t0 = std::chrono::steady_clock::now();
            int tickets[2] = {-1, -1}, sub = 0, col = 0;
            const int nchunks = (total + chunk - 1) / chunk;
            auto drain = [&](int rc_) { for (; col < sub; ++col) (void)f3ds_multi_collect(mg, tickets[col & 1]); return rc_; };      // error path: nothing of this run stays in flight
            while (col < nchunks) {
                if (sub < nchunks && sub - col < 2) {
                    const int k0 = sub * chunk, k = std::min(chunk, total - k0), s = sub & 1;
                    std::vector<const void*> pp; std::vector<size_t> cnt; std::vector<uint32_t*> lp;
                    for (int i = 0; i < k; ++i) { pp.push_back(pts[(size_t)((k0 + i) % distinct)].data()); cnt.push_back(npts); lp.push_back(labels[s][(size_t)i].data()); }
                    const int r = f3ds_multi_submit(mg, pp.data(), cnt.data(), k, &prm, lp.data(), res[s].data(), &tickets[s]);
                    if (r) return drain(r);
                    sub++;
                    continue;
                }
                const int r = f3ds_multi_collect(mg, tickets[col & 1]);
                if (r) { col++; return drain(r); }
                if (last) *last = res[col & 1][0];
                col++;
            }
            *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            return F3DS_OK;
        };
        double sec = 0; f3ds_result r0;
        rc = run(std::min(frames, 2 * chunk), &sec, nullptr);                  // untimed: scratch allocation, code objects
        if (!rc) rc = run(frames, &sec, &r0);
        if (rc) { fprintf(stderr, "--bench: %s %s %s\n", f3ds_strerror(rc), f3ds_last_hip_error(), f3ds_multi_last_error()); f3ds_multi_destroy(mg); return 1; }
        printf("{\"bench\": \"supervoxel_clustering --bench\", \"frames\": %d, \"points_per_frame\": %zu, \"gpus\": %d, \"seconds\": %.4f, \"mpoints_per_s\": %.2f, "
               "\"io\": \"host buffers in, labels in host buffers out\", \"voxels\": %u, \"supervoxels\": %u, \"merges\": %u, \"regions\": %u}\n",
               frames, npts, G, sec, (double)frames * (double)npts / sec / 1e6, r0.n_voxels, r0.n_supervoxels, r0.n_merges, r0.n_regions);
        f3ds_multi_destroy(mg);
        return 0;
    }
    if (gpus > 0) {
        // Multi-GPU batch mode (BASELINE.json config 5): file i -> GPU i mod N, one host thread per GPU, per-point labels
        // gathered on GPU 0 in one RCCL exchange per chunk, then written; chunk k+1 is read and submitted while chunk k is still
        // on the GPUs (f3ds_multi_submit / f3ds_multi_collect).  No ground-truth sweep or scores here.
        if (!thresh_specified || out_labels.empty() || remove_label) { fprintf(stderr, "--gpus needs -t <threshold> and --labels <file>, and does not take -r\n"); return 1; }
        const int per_device = 8;                                  // frames per GPU and chunk (config 5: 64 frames on 8 GPUs)
        f3ds_multi* mg = nullptr;
        int rc = f3ds_multi_create(nullptr, gpus, per_device, &mg);
        if (rc) { fprintf(stderr, "f3ds_multi_create(%d GPUs): %s %s %s\n", gpus, f3ds_strerror(rc), f3ds_last_hip_error(), f3ds_multi_last_error()); return 1; }
        int failed = 0;
        const size_t chunk = (size_t)gpus * per_device;
        struct Chunk { std::vector<size_t> files; std::vector<std::vector<P16>> pts; std::vector<std::vector<uint32_t>> labels; std::vector<size_t> cnt; std::vector<f3ds_result> res; int ticket = -1; };
        auto finish = [&](Chunk& c) {                           // wait for a submitted chunk and write its label files
            if (c.ticket < 0) return;
            const int r = f3ds_multi_collect(mg, c.ticket);
            c.ticket = -1;
            if (r) { fprintf(stderr, "f3ds_multi_segment: %s %s %s\n", f3ds_strerror(r), f3ds_last_hip_error(), f3ds_multi_last_error()); failed = 1; return; }
            for (size_t i = 0; i < c.files.size(); ++i) {
                const std::string& file = file_list[c.files[i]];
                const std::string suffix = file_list.size() > 1 ? "." + std::filesystem::path(file).stem().string() : "";
                FILE* f = fopen((out_labels + suffix).c_str(), "wb");
                if (!f || fwrite(c.labels[i].data(), 4, c.cnt[i], f) != c.cnt[i]) { fprintf(stderr, "writing %s failed\n", (out_labels + suffix).c_str()); failed = 1; }
                if (f) fclose(f);
                const f3ds_result& r2 = c.res[i];
                printf("%s: %llu points, %u voxels, %u supervoxels, %u merges -> %u regions (GPU %d of %d)\n", file.c_str(), (unsigned long long)r2.n_points, r2.n_voxels,
                       r2.n_supervoxels, r2.n_merges, r2.n_regions, f3ds_multi_device_of_frame(mg, (int)i), gpus);
            }
        };
        Chunk ring[2]; int turn = 0;
        for (size_t k0 = 0; k0 < file_list.size(); k0 += chunk, turn ^= 1) {
            Chunk& c = ring[turn];
            finish(c);                                          // (the chunk submitted two rounds ago used these buffers)
            const size_t k1 = std::min(file_list.size(), k0 + chunk);
            c = Chunk();
            for (size_t k = k0; k < k1; ++k) {
                // a file that cannot be read fails the run (exit code 1) and is left out of the batch, as in the single-GPU path's error handling
                size_t n = 0; std::vector<P16> p;
                bool ok = f3ds_pcd_read(file_list[k].c_str(), nullptr, nullptr, 0, &n, nullptr, nullptr) == F3DS_OK;
                if (ok) { p.resize(n); if (n && f3ds_pcd_read(file_list[k].c_str(), p.data(), nullptr, n, &n, nullptr, nullptr) != F3DS_OK) ok = false; }
                if (!ok) { fprintf(stderr, "%s: cannot read PCD file\n", file_list[k].c_str()); failed = 1; continue; }
                p.resize(n);
                c.files.push_back(k); c.pts.push_back(std::move(p)); c.labels.emplace_back(n); c.cnt.push_back(n);
            }
            if (c.files.empty()) continue;
            c.res.resize(c.files.size());
            std::vector<const void*> pp; std::vector<uint32_t*> lp;
            for (size_t i = 0; i < c.files.size(); ++i) { pp.push_back(c.pts[i].data()); lp.push_back(c.labels[i].data()); }
            rc = f3ds_multi_submit(mg, pp.data(), c.cnt.data(), (int)c.files.size(), &prm, lp.data(), c.res.data(), &c.ticket);
            if (rc) { fprintf(stderr, "f3ds_multi_submit: %s %s %s\n", f3ds_strerror(rc), f3ds_last_hip_error(), f3ds_multi_last_error()); failed = 1; break; }
        }
        finish(ring[turn]); finish(ring[turn ^ 1]);
        f3ds_multi_destroy(mg);
        return failed;
    }
    f3ds_ctx* ctx = nullptr;
    int rc = f3ds_create(gpu, &ctx);
    if (rc) { fprintf(stderr, "f3ds_create: %s %s\n", f3ds_strerror(rc), f3ds_last_hip_error()); return 1; }
    std::vector<std::vector<f3ds_performance>> all_performances;
    std::vector<f3ds_performance> best_performances;
    for (const std::string& file : file_list) {
        printf("Loading pointcloud from PCD file '%s'...\n", file.c_str());
        size_t n = 0;
        std::vector<P16> pts; std::vector<uint32_t> gt;
        if (f3ds_pcd_read(file.c_str(), nullptr, nullptr, 0, &n, nullptr, nullptr) == F3DS_OK) {
            pts.resize(n); gt.resize(n);
            if (f3ds_pcd_read(file.c_str(), pts.data(), gt.data(), n, &n, nullptr, nullptr) != F3DS_OK) n = 0;
        }                                                       // like the reference, a failed load leaves an empty cloud (:313)
        pts.resize(n); gt.resize(n);
        if (remove_label) {                                     // :332-336 (has_label is hard-wired true, :309)
            size_t k = 0;
            for (size_t i = 0; i < n; ++i) {
                float z = pts[i].z < 0 ? std::fabs(pts[i].z) : pts[i].z;
                if (gt[i] != (uint32_t)label_to_be_removed && !(z != z)) { pts[k] = pts[i]; gt[k] = gt[i]; ++k; }
            }
            pts.resize(k); gt.resize(k); n = k;
        }
        printf("Pointcloud loaded\nExtracting supervoxels...\n");
        std::vector<uint32_t> labels(n);
        f3ds_result res;
        rc = f3ds_segment(ctx, pts.data(), n, 0, &prm, labels.data(), 0, &res);
        if (rc) { fprintf(stderr, "f3ds_segment: %s %s\n", f3ds_strerror(rc), f3ds_last_hip_error()); f3ds_destroy(ctx); return 1; }
        printf("Found %u supervoxels\nGetting supervoxel adjacency...\n", res.n_supervoxels);
        if (refine_itr > 0) {                                   // :369-375: feeds the viewer only; here the refined labelled voxel cloud can be written
            printf("Refining supervoxels...\n");
            rc = f3ds_refine_supervoxels(ctx, refine_itr);
            size_t nsv = 0, nvx = 0;
            if (!rc) rc = f3ds_get_refined_supervoxels(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &nsv);
            if (!rc) rc = f3ds_get_refined_voxels(ctx, nullptr, nullptr, 0, &nvx);
            if (rc) { fprintf(stderr, "f3ds_refine_supervoxels: %s %s\n", f3ds_strerror(rc), f3ds_last_hip_error()); f3ds_destroy(ctx); return 1; }
            printf("%zu supervoxels after %d refinement iterations\n", nsv, refine_itr);
            if (!out_pcd.empty()) {
                std::vector<float> xyz(nvx * 3); std::vector<uint32_t> col(nvx), lab(nvx); size_t m = 0;
                rc = f3ds_get_voxel_centroid_cloud(ctx, xyz.data(), col.data(), nullptr, nvx, &m);
                if (!rc) rc = f3ds_get_refined_voxels(ctx, lab.data(), nullptr, nvx, &m);
                std::string suffix = file_list.size() > 1 ? "." + std::filesystem::path(file).stem().string() : "";
                if (!rc) rc = f3ds_pcd_write((out_pcd + suffix + ".refined").c_str(), xyz.data(), col.data(), lab.data(), nvx, 1);
                if (rc) { fprintf(stderr, "writing %s.refined: %s\n", out_pcd.c_str(), f3ds_strerror(rc)); f3ds_destroy(ctx); return 1; }
            }
        }
        printf("Segmentation initialization...\n");
        if (ml || al) DEBUG("Lambda: %f\n", res.lambda);
        if (!thresh_specified) {                                // all_thresh + best_thresh (:428-437)
            std::vector<float> ts(4096); std::vector<f3ds_performance> ps(4096);
            size_t nt = 0; float best_t = 0; f3ds_performance best_p;
            rc = f3ds_auto_threshold(ctx, &prm, gt.data(), 0.8f, 1.0f, 0.005f, ts.data(), ps.data(), ts.size(), &nt, &best_t, &best_p, labels.data(), 0, &res);
            if (rc) { fprintf(stderr, "f3ds_auto_threshold: %s %s\n", f3ds_strerror(rc), f3ds_last_hip_error()); f3ds_destroy(ctx); return 1; }
            ps.resize(nt < ps.size() ? nt : ps.size());
            all_performances.push_back(ps);
            printf("Using best threshold: %f (F-score %f, voi %f)\n", best_t, best_p.fscore, best_p.voi);
            prm.threshold = best_t;
        }
        printf("Initialization complete\nStarting clustering...\nClustering complete\n");
        printf("Initializing testing suite...\n");
        f3ds_performance score;
        rc = f3ds_evaluate(ctx, gt.data(), &score);             // Testing(get_labeled_cloud(), truth).eval_performance() (:462-463)
        if (rc) { fprintf(stderr, "f3ds_evaluate: %s %s\n", f3ds_strerror(rc), f3ds_last_hip_error()); f3ds_destroy(ctx); return 1; }
        best_performances.push_back(score);
        printf("%llu points, %u voxels, %u supervoxels, %u adjacencies, %u merges -> %u regions (%.3f ms on GPU %d)\n",
               (unsigned long long)res.n_points, res.n_voxels, res.n_supervoxels, res.n_edges, res.n_merges, res.n_regions, res.ms_total, gpu);
        if (!out_pcd.empty() || !out_labels.empty()) {
            std::string suffix = file_list.size() > 1 ? "." + std::filesystem::path(file).stem().string() : "";
            if (!out_pcd.empty()) {
                size_t nv = 0;
                f3ds_get_voxel_cloud(ctx, nullptr, nullptr, nullptr, 0, &nv);
                std::vector<float> xyz(nv * 3); std::vector<uint32_t> lab(nv), col(nv);
                rc = f3ds_get_voxel_cloud(ctx, xyz.data(), lab.data(), col.data(), nv, &nv);
                if (!rc) rc = f3ds_pcd_write((out_pcd + suffix).c_str(), xyz.data(), col.data(), lab.data(), nv, 1);
                if (rc) { fprintf(stderr, "writing %s: %s\n", out_pcd.c_str(), f3ds_strerror(rc)); f3ds_destroy(ctx); return 1; }
            }
            if (!out_labels.empty()) {
                FILE* f = fopen((out_labels + suffix).c_str(), "wb");
                if (!f || fwrite(labels.data(), 4, n, f) != n) { fprintf(stderr, "writing %s failed\n", out_labels.c_str()); if (f) fclose(f); f3ds_destroy(ctx); return 1; }
                fclose(f);
            }
        }
        if (!dump_dir.empty()) {
            // Headless replacement for visualize() (:586-700): the clouds and the graph the viewer would show, as files.
            //   voxel_centroids.pcd     getVoxelCentroidCloud() (:359)                     "voxel centroids"
            //   colored_voxels.pcd      Clustering::get_colored_cloud() + region label (:443)  "colored voxels"
            //   supervoxel_normals.pcd  makeSupervoxelNormalCloud(refined clusters) (:373)  "supervoxel_normals"
            //   adjacency.csv           get_currentstate().second (:444) with the centroids of supervoxel_clusters.at(label) (:636-672)
            std::error_code ec;
            const std::string dir = file_list.size() > 1 ? dump_dir + "/" + std::filesystem::path(file).stem().string() : dump_dir;
            std::filesystem::create_directories(dir, ec);
            size_t nv = 0, nc = 0, ns = 0, nr = 0, ne = 0;
            rc = f3ds_get_voxel_centroid_cloud(ctx, nullptr, nullptr, nullptr, 0, &nv);
            std::vector<float> vxyz(nv * 3); std::vector<uint32_t> vcol(nv), vsv(nv);
            if (!rc) rc = f3ds_get_voxel_centroid_cloud(ctx, vxyz.data(), vcol.data(), vsv.data(), nv, &nv);
            if (!rc) rc = f3ds_pcd_write((dir + "/voxel_centroids.pcd").c_str(), vxyz.data(), vcol.data(), vsv.data(), nv, 1);
            if (!rc) rc = f3ds_get_voxel_cloud(ctx, nullptr, nullptr, nullptr, 0, &nc);
            std::vector<float> cxyz(nc * 3); std::vector<uint32_t> clab(nc), ccol(nc);
            if (!rc) rc = f3ds_get_voxel_cloud(ctx, cxyz.data(), clab.data(), ccol.data(), nc, &nc);
            if (!rc) rc = f3ds_pcd_write((dir + "/colored_voxels.pcd").c_str(), cxyz.data(), ccol.data(), clab.data(), nc, 1);
            if (!rc) rc = f3ds_get_refined_supervoxels(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &nr);
            std::vector<uint32_t> rlab(nr), rcnt(nr); std::vector<float> rxyz(nr * 3), rrgb(nr * 3), rnrm(nr * 3);
            if (!rc) rc = f3ds_get_refined_supervoxels(ctx, rlab.data(), rxyz.data(), rrgb.data(), rnrm.data(), rcnt.data(), nr, &nr);
            if (!rc && !write_normal_pcd(dir + "/supervoxel_normals.pcd", rxyz, rnrm, nr)) rc = F3DS_ERR_IO;
            if (!rc) rc = f3ds_get_supervoxels(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &ns);
            std::vector<uint32_t> slab(ns), scnt(ns); std::vector<float> sxyz(ns * 3), srgb(ns * 3), snrm(ns * 3);
            if (!rc) rc = f3ds_get_supervoxels(ctx, slab.data(), sxyz.data(), srgb.data(), snrm.data(), scnt.data(), ns, &ns);
            if (!rc) rc = f3ds_get_region_adjacency(ctx, nullptr, 0, &ne);
            std::vector<uint32_t> pairs(ne * 2);
            if (!rc) rc = f3ds_get_region_adjacency(ctx, pairs.data(), ne, &ne);
            if (!rc) {
                FILE* f = fopen((dir + "/adjacency.csv").c_str(), "w");
                if (!f) rc = F3DS_ERR_IO;
                else {
                    fprintf(f, "label_a,label_b,ax,ay,az,bx,by,bz\n");
                    for (size_t e = 0; e < ne; ++e) {
                        const size_t ia = std::lower_bound(slab.begin(), slab.end(), pairs[2 * e]) - slab.begin(), ib = std::lower_bound(slab.begin(), slab.end(), pairs[2 * e + 1]) - slab.begin();
                        if (ia >= ns || ib >= ns) continue;
                        fprintf(f, "%u,%u,%.9g,%.9g,%.9g,%.9g,%.9g,%.9g\n", pairs[2 * e], pairs[2 * e + 1], sxyz[3 * ia], sxyz[3 * ia + 1], sxyz[3 * ia + 2], sxyz[3 * ib], sxyz[3 * ib + 1], sxyz[3 * ib + 2]);
                    }
                    fclose(f);
                }
            }
            if (rc) { fprintf(stderr, "--dump %s: %s %s\n", dir.c_str(), f3ds_strerror(rc), f3ds_last_hip_error()); f3ds_destroy(ctx); return 1; }
            printf("Dumped %zu voxel centroids, %zu coloured voxels, %zu refined supervoxel normals, %zu region adjacencies to %s\n", nv, nc, nr, ne, dir.c_str());
        }
    }
    f3ds_destroy(ctx);
    // manageAllPerformances (:475-518): one line per file, one "value;" per threshold
    {
        const char* names[7] = {"voi", "precision", "recall", "fscore", "wov", "fpr", "fnr"};
        for (int k = 0; k < 7; ++k) {
            FILE* f = fopen((test_filename + "_" + names[k] + ".csv").c_str(), "w");
            if (!f) continue;
            for (const auto& row : all_performances) {
                for (const f3ds_performance& p : row) fprintf(f, "%g;", (&p.voi)[k]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    // printBestPerformances (:520-557); its running mean uses the integer 1/count, so with several files
    // the "average" it prints is the first file's scores -- kept as is
    if (!best_performances.empty()) {
        const f3ds_performance& p = best_performances.size() == 1 ? best_performances.back() : best_performances.front();
        printf("%s:\nVOI\t%f\nPrec.\t%f\nRecall\t%f\nF-score\t%f\nWOv\t%f\nFPR\t%f\nFNR\t%f\n",
               best_performances.size() == 1 ? "Scores" : "Average scores", p.voi, p.precision, p.recall, p.fscore, p.wov, p.fpr, p.fnr);
    }
    return 0;
}
