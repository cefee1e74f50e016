// f3ds_multi.cpp -- multi-GPU batch driver in one process (BASELINE.json config 5 / north_star: "independent frames shard
// one-per-GPU across the 8 x MI355X node with a single RCCL gather over xGMI for the label output").
//
// The reference has no counterpart (single-threaded CLI, /root/reference/CMakeLists.txt:5); this is the C++ side of what
// fast-3d-pointcloud-segmentation_amd/batch.py does with one process per GPU under torch.distributed.
//
//   * frame i runs on devices[i mod G]; one host thread per GPU drives f3ds_segment_batch on that GPU's frames
//     (host buffers in, labels left in that GPU's memory);
//   * the label output goes to devices[0] in ONE grouped RCCL exchange: every other GPU ncclSend()s its label block, GPU 0
//     posts the matching ncclRecv()s, all between ncclGroupStart / ncclGroupEnd (frames are ragged, so this is the
//     send/recv form of a gather; with equal frames it moves exactly what ncclGather would).  Each peer uses its own xGMI
//     link to GPU 0; nothing else crosses GPUs.  GPU 0's own frames are written straight into the gathered block;
//   * librccl is loaded at run time (dlopen): libf3ds itself has no link-time dependency on it, and a process that already
//     carries a librccl (PyTorch bundles one) keeps using that copy.  With one GPU no RCCL call is made at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/f3ds.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (lib) return true;
        const char* names[] = {getenv("F3DS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
#define F3DS_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); if (!field) { dlclose(lib); lib = nullptr; return false; }
        F3DS_SYM(CommInitAll, "ncclCommInitAll") F3DS_SYM(CommDestroy, "ncclCommDestroy") F3DS_SYM(GroupStart, "ncclGroupStart") F3DS_SYM(GroupEnd, "ncclGroupEnd")
        F3DS_SYM(Send, "ncclSend") F3DS_SYM(Recv, "ncclRecv") F3DS_SYM(GetErrorString, "ncclGetErrorString")
#undef F3DS_SYM
        return true;
    }
};
Rccl g_rccl;
thread_local std::string g_multi_error;

struct PerDevice {
    int device = 0;
    std::vector<f3ds_ctx*> ctxs;
    hipStream_t stream = nullptr;          // gather / copy-out stream of this device
    uint32_t* labels = nullptr;            // this device's label block (device 0: the gathered block of all devices)
    size_t labels_cap = 0;                 // in uint32
    uint32_t* loop = nullptr;              // F3DS_MULTI_FORCE_RCCL with one device: the block after a send/recv to itself
    size_t loop_cap = 0;
    int rc = 0;
    std::string err;
};

}  // namespace

struct f3ds_multi {
    std::vector<PerDevice> dev;
    std::vector<ncclComm_t> comm;          // one per device (ncclCommInitAll), empty with a single device
    int max_frames_per_device = 0;
};

extern "C" {

const char* f3ds_multi_last_error(void) { return g_multi_error.c_str(); }

void f3ds_multi_destroy(f3ds_multi* m) {
    if (!m) return;
    for (size_t d = 0; d < m->comm.size(); ++d) if (m->comm[d]) { (void)hipSetDevice(m->dev[d].device); (void)g_rccl.CommDestroy(m->comm[d]); }
    for (PerDevice& p : m->dev) {
        (void)hipSetDevice(p.device);
        for (f3ds_ctx* c : p.ctxs) f3ds_destroy(c);
        if (p.labels) (void)hipFree(p.labels);
        if (p.loop) (void)hipFree(p.loop);
        if (p.stream) (void)hipStreamDestroy(p.stream);
    }
    delete m;
}

int f3ds_multi_create(const int* devices, int n_devices, int max_frames_per_device, f3ds_multi** out) {
    if (!out) return F3DS_ERR_ARG;
    *out = nullptr;
    const int visible = f3ds_device_count();
    if (visible <= 0) return F3DS_ERR_NO_DEVICE;
    if (n_devices <= 0 || n_devices > visible || max_frames_per_device <= 0) return F3DS_ERR_ARG;
    f3ds_multi* m = new f3ds_multi;
    m->max_frames_per_device = max_frames_per_device;
    m->dev.resize((size_t)n_devices);
    for (int d = 0; d < n_devices; ++d) {
        const int id = devices ? devices[d] : d;
        if (id < 0 || id >= visible) { f3ds_multi_destroy(m); return F3DS_ERR_ARG; }
        for (int e = 0; e < d; ++e) if (m->dev[(size_t)e].device == id) { f3ds_multi_destroy(m); return F3DS_ERR_ARG; }
        m->dev[(size_t)d].device = id;
        if (hipSetDevice(id) != hipSuccess || hipStreamCreateWithFlags(&m->dev[(size_t)d].stream, hipStreamNonBlocking) != hipSuccess) { f3ds_multi_destroy(m); return F3DS_ERR_HIP; }
    }
    // development / single-GPU test boxes: F3DS_MULTI_FORCE_RCCL=1 builds the communicator with one device too and sends the
    // label block through RCCL to itself, so that the library loading and the grouped send/recv are exercised on one GPU
    if (n_devices > 1 || getenv("F3DS_MULTI_FORCE_RCCL")) {
        if (!g_rccl.load()) { g_multi_error = "librccl not found (F3DS_RCCL_LIB, librccl.so.1, /opt/rocm/lib)"; f3ds_multi_destroy(m); return F3DS_ERR_UNSUPPORTED; }
        std::vector<int> ids;
        for (const PerDevice& p : m->dev) ids.push_back(p.device);
        m->comm.assign((size_t)n_devices, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), n_devices, ids.data());
        if (r != ncclSuccess) { g_multi_error = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r); m->comm.clear(); f3ds_multi_destroy(m); return F3DS_ERR_HIP; }
    }
    *out = m;
    return F3DS_OK;
}

int f3ds_multi_devices(const f3ds_multi* m) { return m ? (int)m->dev.size() : 0; }

// device index (position in the devices array) a frame runs on, and its position among that device's frames
int f3ds_multi_device_of_frame(const f3ds_multi* m, int frame) { return m && frame >= 0 ? frame % (int)m->dev.size() : -1; }

int f3ds_multi_segment(f3ds_multi* m, const void* const* points, const size_t* counts, int n_frames, const f3ds_params* params,
                       uint32_t* const* point_labels, f3ds_result* results) {
    if (!m || !points || !counts || !params || n_frames < 0) return F3DS_ERR_ARG;
    const int G = (int)m->dev.size();
    if (n_frames > G * m->max_frames_per_device) return F3DS_ERR_CAPACITY;
    if (n_frames == 0) return F3DS_OK;
    // frames of every device, in frame order; label offsets inside the device's block
    std::vector<std::vector<int>> mine((size_t)G);
    for (int i = 0; i < n_frames; ++i) mine[(size_t)(i % G)].push_back(i);
    std::vector<size_t> block((size_t)G, 0), base((size_t)G, 0), off((size_t)n_frames, 0);
    for (int d = 0; d < G; ++d) for (int i : mine[(size_t)d]) { off[(size_t)i] = block[(size_t)d]; block[(size_t)d] += counts[i]; }
    size_t total = 0;
    for (int d = 0; d < G; ++d) { base[(size_t)d] = total; total += block[(size_t)d]; }       // device 0's gathered block: [dev 0 | dev 1 | ...]
    // ---- one host thread per GPU: its frames as one f3ds_segment_batch, labels into its device block
    auto work = [&](int d) {
        PerDevice& p = m->dev[(size_t)d];
        p.rc = 0; p.err.clear();
        const std::vector<int>& fr = mine[(size_t)d];
        if (hipSetDevice(p.device) != hipSuccess) { p.rc = F3DS_ERR_HIP; return; }
        const size_t need = d == 0 ? total : block[(size_t)d];
        if (p.labels_cap < need) {
            if (p.labels) (void)hipFree(p.labels);
            p.labels = nullptr; p.labels_cap = 0;
            if (hipMalloc((void**)&p.labels, (need + need / 4 + 64) * sizeof(uint32_t)) != hipSuccess) { p.rc = F3DS_ERR_HIP; p.err = "hipMalloc(label block)"; return; }
            p.labels_cap = need + need / 4 + 64;
        }
        if (fr.empty()) return;
        while (p.ctxs.size() < fr.size()) { f3ds_ctx* c = nullptr; const int rc = f3ds_create(p.device, &c); if (rc) { p.rc = rc; p.err = f3ds_last_hip_error(); return; } p.ctxs.push_back(c); }
        std::vector<const void*> pp; std::vector<size_t> cnt; std::vector<uint32_t*> lp; std::vector<f3ds_result> res(fr.size());
        for (int i : fr) { pp.push_back(points[i]); cnt.push_back(counts[i]); lp.push_back(p.labels + (d == 0 ? base[0] : 0) + off[(size_t)i]); }
        // host points in, device labels out: f3ds_segment_batch takes one flag per side
        p.rc = f3ds_segment_batch(p.ctxs.data(), (int)fr.size(), pp.data(), cnt.data(), 0, params, lp.data(), 1, res.data());
        if (p.rc) { p.err = f3ds_last_hip_error(); return; }
        if (results) for (size_t k = 0; k < fr.size(); ++k) results[fr[k]] = res[k];
    };
    std::vector<std::thread> th;
    for (int d = 1; d < G; ++d) th.emplace_back(work, d);
    work(0);
    for (std::thread& t : th) t.join();
    for (int d = 0; d < G; ++d) if (m->dev[(size_t)d].rc) { g_multi_error = "device " + std::to_string(m->dev[(size_t)d].device) + ": " + m->dev[(size_t)d].err; return m->dev[(size_t)d].rc; }
    // ---- label output: every peer's block to device 0 in one grouped exchange (f3ds_segment_batch has returned: the blocks are complete)
    PerDevice& root = m->dev[0];
    if (G > 1) {
        ncclResult_t r = g_rccl.GroupStart();
        for (int d = 1; d < G && r == ncclSuccess; ++d) {
            if (!block[(size_t)d]) continue;
            (void)hipSetDevice(m->dev[(size_t)d].device);
            r = g_rccl.Send(m->dev[(size_t)d].labels, block[(size_t)d], ncclUint32, 0, m->comm[(size_t)d], m->dev[(size_t)d].stream);
            if (r != ncclSuccess) break;
            (void)hipSetDevice(root.device);
            r = g_rccl.Recv(root.labels + base[(size_t)d], block[(size_t)d], ncclUint32, d, m->comm[0], root.stream);
        }
        const ncclResult_t e = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) { g_multi_error = std::string("RCCL label exchange: ") + g_rccl.GetErrorString(r); return F3DS_ERR_HIP; }
        for (int d = 1; d < G; ++d) { (void)hipSetDevice(m->dev[(size_t)d].device); if (hipStreamSynchronize(m->dev[(size_t)d].stream) != hipSuccess) return F3DS_ERR_HIP; }
    }
    // ---- gathered labels (device 0) -> the caller's host buffers
    if (hipSetDevice(root.device) != hipSuccess) return F3DS_ERR_HIP;
    const uint32_t* gathered = root.labels;
    if (G == 1 && !m->comm.empty() && total) {       // forced RCCL path on one GPU: the block goes through a send/recv pair to itself
        if (root.loop_cap < total) {
            if (root.loop) (void)hipFree(root.loop);
            root.loop = nullptr; root.loop_cap = 0;
            if (hipMalloc((void**)&root.loop, total * sizeof(uint32_t)) != hipSuccess) return F3DS_ERR_HIP;
            root.loop_cap = total;
        }
        ncclResult_t r = g_rccl.GroupStart();
        if (r == ncclSuccess) r = g_rccl.Send(root.labels, total, ncclUint32, 0, m->comm[0], root.stream);
        if (r == ncclSuccess) r = g_rccl.Recv(root.loop, total, ncclUint32, 0, m->comm[0], root.stream);
        const ncclResult_t e = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) { g_multi_error = std::string("RCCL self exchange: ") + g_rccl.GetErrorString(r); return F3DS_ERR_HIP; }
        gathered = root.loop;
    }
    if (point_labels)
        for (int i = 0; i < n_frames; ++i)
            if (point_labels[i] && counts[i] && hipMemcpyAsync(point_labels[i], gathered + base[(size_t)(i % G)] + off[(size_t)i], counts[i] * sizeof(uint32_t), hipMemcpyDeviceToHost, root.stream) != hipSuccess)
                return F3DS_ERR_HIP;
    if (hipStreamSynchronize(root.stream) != hipSuccess) return F3DS_ERR_HIP;
    return F3DS_OK;
}

// the gathered label block on device 0 after f3ds_multi_segment (device pointer, uint32 per point; frames of device d at
// their running offsets inside [dev 0 | dev 1 | ...]) -- for callers that keep the labels on the GPU
const uint32_t* f3ds_multi_gathered_labels(const f3ds_multi* m) { return m && !m->dev.empty() ? m->dev[0].labels : nullptr; }

}  // extern "C"
