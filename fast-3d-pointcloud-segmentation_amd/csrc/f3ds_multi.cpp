// f3ds_multi.cpp -- multi-GPU batch driver in one process (BASELINE.json config 5 / north_star: "independent frames shard
// one-per-GPU across the 8 x MI355X node with a single RCCL gather over xGMI for the label output").
//
// The reference has no counterpart (single-threaded CLI, /root/reference/CMakeLists.txt:5); this is the C++ side of what
// fast-3d-pointcloud-segmentation_amd/batch.py does with one process per GPU under torch.distributed.
//
//   * frame i of a batch runs on devices[i mod G]; ONE PERSISTENT HOST THREAD PER GPU runs that GPU's frames of a batch as one
//     f3ds_segment_batch (host buffers in, labels left in that GPU's memory);
//   * the label output of a batch goes to devices[0] in ONE grouped RCCL exchange: every other GPU ncclSend()s its label block,
//     GPU 0 posts the matching ncclRecv()s, all between ncclGroupStart / ncclGroupEnd (frames are ragged, so this is the
//     send/recv form of a gather; with equal frames it moves exactly what ncclGather would).  Each peer uses its own xGMI link
//     to GPU 0; nothing else crosses GPUs.  GPU 0's own frames are written straight into the gathered block;
//   * batches are PIPELINED: f3ds_multi_submit() queues a batch and returns; an exchange thread gathers batch k and copies its
//     labels to the caller's host buffers while the GPU threads already compute batch k+1 (two label-block slots per device);
//     f3ds_multi_collect() waits for a batch.  f3ds_multi_segment() = submit + collect;
//   * label blocks are allocated in f3ds_multi_create / f3ds_multi_reserve, outside the steady state (hipFree stalls the device);
//   * librccl is loaded at run time (dlopen, once per process): libf3ds itself has no link-time dependency on it, and a process
//     that already carries a librccl (PyTorch bundles one) keeps using that copy.  With one GPU no RCCL call is made at all;
//   * an RCCL error inside the group aborts the communicators (ncclCommAbort) instead of closing a group with an unmatched send --
//     the driver then refuses further batches (F3DS_ERR_HIP) rather than hang;
//   * the calling thread's current HIP device is left as it was;
//   * LOGICAL DEVICES (F3DS_MULTI_LOGICAL=1, tests on 1-GPU boxes): the `devices` array may then name one GPU several times (or more
//     entries than GPUs are visible; NULL = logical device d on GPU d mod visible).  Every entry is a device of its own to the driver --
//     its own worker thread, contexts, label blocks, stream, its own block/base/off arithmetic -- so everything of the G > 1 path runs
//     except the wire: RCCL refuses a communicator with duplicate GPUs, so the peers' blocks reach devices[0]'s gathered block with
//     hipMemcpyAsync (device to device) where a real node uses ncclSend / ncclRecv.  Never a production mode: no speed-up comes of it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/f3ds.h"
#include "f3ds_dev.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::once_flag once;
    bool ok = false;
    bool load() {       // concurrent f3ds_multi_create calls: the library is opened once
        std::call_once(once, [this] { ok = load_once(); });
        return ok;
    }
    bool load_once() {
        const char* names[] = {getenv("F3DS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
#define F3DS_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); if (!field) { dlclose(lib); lib = nullptr; return false; }
        F3DS_SYM(CommInitAll, "ncclCommInitAll") F3DS_SYM(CommDestroy, "ncclCommDestroy") F3DS_SYM(CommAbort, "ncclCommAbort") F3DS_SYM(GroupStart, "ncclGroupStart")
        F3DS_SYM(GroupEnd, "ncclGroupEnd") F3DS_SYM(Send, "ncclSend") F3DS_SYM(Recv, "ncclRecv") F3DS_SYM(GetErrorString, "ncclGetErrorString")
#undef F3DS_SYM
        return true;
    }
};
Rccl g_rccl;
thread_local std::string g_multi_error;

// the calling thread's HIP device is restored when an entry point returns
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

constexpr int SLOTS = 2;       // batches in flight: one computing, one being gathered / copied out

// one submitted batch
struct Job {
    int ticket = -1, slot = 0, n_frames = 0;
    std::vector<const void*> points; std::vector<size_t> counts; std::vector<uint32_t*> labels;
    f3ds_params params; f3ds_result* results = nullptr;
    std::vector<std::vector<int>> mine;                  // frames of every device, in frame order
    std::vector<size_t> block, base, off; size_t total = 0;       // label counts per device, offsets inside device 0's gathered block, per frame inside its device's block
    std::atomic<int> devices_left{0};
    int rc = 0; std::string err;
    bool done = false, collected = false;
};

struct PerDevice {
    int device = 0;
    std::vector<f3ds_ctx*> ctxs;
    hipStream_t stream = nullptr;          // gather / copy-out stream of this device
    uint32_t* labels[SLOTS] = {};          // this device's label block per slot (device 0: the gathered block of all devices)
    size_t labels_cap[SLOTS] = {};         // in uint32
    uint32_t* loop = nullptr;              // F3DS_MULTI_FORCE_RCCL with one device: the block after a send/recv to itself
    size_t loop_cap = 0;
    std::thread worker;
    std::deque<Job*> queue;                // jobs this device still has to compute (guarded by f3ds_multi::mu)
};

}  // namespace

struct f3ds_multi {
    std::vector<PerDevice> dev;
    std::vector<ncclComm_t> comm;          // one per device (ncclCommInitAll), empty with a single device
    int max_frames_per_device = 0;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job*> exch_queue;           // batches whose compute is complete on every device, in submission order
    std::thread exchanger;
    std::unique_ptr<Job> ring[SLOTS];
    int next_ticket = 0;
    std::atomic<int> last_slot{-1};        // written by the exchange thread, read by f3ds_multi_gathered_labels without the lock
    bool stop = false;
    std::atomic<bool> broken{false};       // the communicators were aborted after an RCCL error (set by the exchange thread)
    bool logical = false;                  // F3DS_MULTI_LOGICAL: entries of `devices` may share a GPU; the exchange is a device-to-device copy
};

namespace {

int grow_block(uint32_t** p, size_t* cap, size_t need) {
    if (*cap >= need) return F3DS_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = need + need / 4 + 64;
    if (hipMalloc((void**)p, want * sizeof(uint32_t)) != hipSuccess) return F3DS_ERR_HIP;
    *cap = want;
    return F3DS_OK;
}

// a device's share of a batch: its frames as one f3ds_segment_batch, labels into its block of the batch's slot
void compute(f3ds_multi* m, int d, Job& j, int* rc_out, std::string* err_out) {
    PerDevice& p = m->dev[(size_t)d];
    const std::vector<int>& fr = j.mine[(size_t)d];
    if (hipSetDevice(p.device) != hipSuccess) { *rc_out = F3DS_ERR_HIP; *err_out = "hipSetDevice"; return; }
    const size_t need = d == 0 ? j.total : j.block[(size_t)d];
    // (a no-op after f3ds_multi_reserve or the first batch of this size: the steady state does not allocate)
    if (grow_block(&p.labels[j.slot], &p.labels_cap[j.slot], need)) { *rc_out = F3DS_ERR_HIP; *err_out = "hipMalloc(label block)"; return; }
    if (fr.empty()) return;
    while (p.ctxs.size() < fr.size()) { f3ds_ctx* c = nullptr; const int rc = f3ds_create(p.device, &c); if (rc) { *rc_out = rc; *err_out = f3ds_last_hip_error(); return; } p.ctxs.push_back(c); }
    std::vector<const void*> pp; std::vector<size_t> cnt; std::vector<uint32_t*> lp; std::vector<f3ds_result> res(fr.size());
    for (int i : fr) { pp.push_back(j.points[(size_t)i]); cnt.push_back(j.counts[(size_t)i]); lp.push_back(p.labels[j.slot] + (d == 0 ? j.base[0] : 0) + j.off[(size_t)i]); }
    // host points in, device labels out: f3ds_segment_batch takes one flag per side
    const int rc = f3ds_segment_batch(p.ctxs.data(), (int)fr.size(), pp.data(), cnt.data(), 0, &j.params, lp.data(), 1, res.data());
    if (rc) { *rc_out = rc; *err_out = f3ds_last_hip_error(); return; }
    if (j.results) for (size_t k = 0; k < fr.size(); ++k) j.results[fr[k]] = res[k];
}

void worker_loop(f3ds_multi* m, int d) {
    for (;;) {
        Job* j = nullptr;
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv_work.wait(lk, [&] { return m->stop || !m->dev[(size_t)d].queue.empty(); });
            if (m->dev[(size_t)d].queue.empty()) return;
            j = m->dev[(size_t)d].queue.front(); m->dev[(size_t)d].queue.pop_front();
        }
        int rc = 0; std::string err;
        compute(m, d, *j, &rc, &err);
        {
            std::lock_guard<std::mutex> lk(m->mu);
            if (rc && !j->rc) { j->rc = rc; j->err = "device " + std::to_string(m->dev[(size_t)d].device) + ": " + err; }
            // every device runs its batches in submission order, so "last device done" happens in submission order too
            if (j->devices_left.fetch_sub(1) == 1) { m->exch_queue.push_back(j); m->cv_work.notify_all(); }
        }
    }
}

// label output of one batch: every peer's block to device 0 in one grouped exchange, then device 0 -> the caller's host buffers
int exchange(f3ds_multi* m, Job& j, std::string* err) {
    const int G = (int)m->dev.size();
    PerDevice& root = m->dev[0];
    const int s = j.slot;
    if (G > 1 && m->logical) {
        // logical devices: same arithmetic (block / base per device), the peers' blocks copied device to device on the root's stream
        for (int d = 1; d < G; ++d) {
            if (!j.block[(size_t)d]) continue;
            if (hipSetDevice(root.device) != hipSuccess ||
                hipMemcpyAsync(root.labels[s] + j.base[(size_t)d], m->dev[(size_t)d].labels[s], j.block[(size_t)d] * sizeof(uint32_t), hipMemcpyDeviceToDevice, root.stream) != hipSuccess) {
                *err = "hipMemcpyAsync(logical peer block)"; return F3DS_ERR_HIP;
            }
        }
    } else if (G > 1) {
        ncclResult_t r = g_rccl.GroupStart();
        bool posted = false;
        for (int d = 1; d < G && r == ncclSuccess; ++d) {
            if (!j.block[(size_t)d]) continue;
            (void)hipSetDevice(m->dev[(size_t)d].device);
            r = g_rccl.Send(m->dev[(size_t)d].labels[s], j.block[(size_t)d], ncclUint32, 0, m->comm[(size_t)d], m->dev[(size_t)d].stream);
            posted = true;
            if (r != ncclSuccess) break;
            (void)hipSetDevice(root.device);
            r = g_rccl.Recv(root.labels[s] + j.base[(size_t)d], j.block[(size_t)d], ncclUint32, d, m->comm[0], root.stream);
        }
        if (r != ncclSuccess && posted) {
            // a send without its receive (or the reverse) is pending inside the open group: closing the group could wait for ever.
            // Abort every communicator; this driver does no further exchange.
            for (size_t d = 0; d < m->comm.size(); ++d) if (m->comm[d]) { (void)hipSetDevice(m->dev[d].device); (void)g_rccl.CommAbort(m->comm[d]); m->comm[d] = nullptr; }
            m->broken.store(true);
            *err = std::string("RCCL label exchange (communicators aborted): ") + g_rccl.GetErrorString(r);
            return F3DS_ERR_HIP;
        }
        const ncclResult_t e = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) { *err = std::string("RCCL label exchange: ") + g_rccl.GetErrorString(r); return F3DS_ERR_HIP; }
        for (int d = 1; d < G; ++d) { (void)hipSetDevice(m->dev[(size_t)d].device); if (hipStreamSynchronize(m->dev[(size_t)d].stream) != hipSuccess) { *err = "hipStreamSynchronize(peer)"; return F3DS_ERR_HIP; } }
    }
    if (hipSetDevice(root.device) != hipSuccess) { *err = "hipSetDevice"; return F3DS_ERR_HIP; }
    const uint32_t* gathered = root.labels[s];
    if (G == 1 && !m->comm.empty() && j.total) {       // forced RCCL path on one GPU: the block goes through a send/recv pair to itself
        if (grow_block(&root.loop, &root.loop_cap, j.total)) { *err = "hipMalloc(loop block)"; return F3DS_ERR_HIP; }
        ncclResult_t r = g_rccl.GroupStart();
        if (r == ncclSuccess) r = g_rccl.Send(root.labels[s], j.total, ncclUint32, 0, m->comm[0], root.stream);
        if (r == ncclSuccess) r = g_rccl.Recv(root.loop, j.total, ncclUint32, 0, m->comm[0], root.stream);
        const ncclResult_t e = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) { *err = std::string("RCCL self exchange: ") + g_rccl.GetErrorString(r); return F3DS_ERR_HIP; }
        gathered = root.loop;
    }
    for (int i = 0; i < j.n_frames; ++i)
        if (j.labels[(size_t)i] && j.counts[(size_t)i] &&
            hipMemcpyAsync(j.labels[(size_t)i], gathered + j.base[(size_t)(i % G)] + j.off[(size_t)i], j.counts[(size_t)i] * sizeof(uint32_t), hipMemcpyDeviceToHost, root.stream) != hipSuccess) {
            *err = "hipMemcpyAsync(labels to host)"; return F3DS_ERR_HIP;
        }
    if (hipStreamSynchronize(root.stream) != hipSuccess) { *err = "hipStreamSynchronize(root)"; return F3DS_ERR_HIP; }
    return F3DS_OK;
}

void exchanger_loop(f3ds_multi* m) {
    for (;;) {
        Job* j = nullptr;
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv_work.wait(lk, [&] { return m->stop || !m->exch_queue.empty(); });
            if (m->exch_queue.empty()) return;
            j = m->exch_queue.front(); m->exch_queue.pop_front();
        }
        int rc = j->rc; std::string err;
        if (!rc) rc = exchange(m, *j, &err);
        {
            std::lock_guard<std::mutex> lk(m->mu);
            if (rc && !j->rc) { j->rc = rc; j->err = err; }
            j->done = true; m->last_slot.store(j->slot);
        }
        m->cv_done.notify_all();
    }
}

}  // namespace

extern "C" {

const char* f3ds_multi_last_error(void) { return g_multi_error.c_str(); }

void f3ds_multi_destroy(f3ds_multi* m) {
    if (!m) return;
    DeviceGuard guard;
    {
        std::unique_lock<std::mutex> lk(m->mu);
        // batches still in flight complete first (their host buffers belong to the caller)
        m->cv_done.wait(lk, [&] { for (auto& j : m->ring) if (j && !j->done) return false; return true; });
        m->stop = true;
    }
    m->cv_work.notify_all();
    for (PerDevice& p : m->dev) if (p.worker.joinable()) p.worker.join();
    if (m->exchanger.joinable()) m->exchanger.join();
    for (size_t d = 0; d < m->comm.size(); ++d) if (m->comm[d]) { (void)hipSetDevice(m->dev[d].device); (void)g_rccl.CommDestroy(m->comm[d]); }
    for (PerDevice& p : m->dev) {
        (void)hipSetDevice(p.device);
        for (f3ds_ctx* c : p.ctxs) f3ds_destroy(c);
        for (int s = 0; s < SLOTS; ++s) if (p.labels[s]) (void)hipFree(p.labels[s]);
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
    const bool logical = f3ds::dev_getenv("F3DS_MULTI_LOGICAL") != nullptr;      // tests: several logical devices on one GPU (see the head of this file)
    if (n_devices <= 0 || (n_devices > visible && !logical) || n_devices > 64 || max_frames_per_device <= 0) return F3DS_ERR_ARG;
    DeviceGuard guard;
    f3ds_multi* m = new f3ds_multi;
    m->max_frames_per_device = max_frames_per_device;
    m->logical = logical;
    m->dev.resize((size_t)n_devices);
    for (int d = 0; d < n_devices; ++d) {
        const int id = devices ? devices[d] : (logical ? d % visible : d);
        if (id < 0 || id >= visible) { f3ds_multi_destroy(m); return F3DS_ERR_ARG; }
        if (!logical) for (int e = 0; e < d; ++e) if (m->dev[(size_t)e].device == id) { f3ds_multi_destroy(m); return F3DS_ERR_ARG; }
        m->dev[(size_t)d].device = id;
        if (hipSetDevice(id) != hipSuccess || hipStreamCreateWithFlags(&m->dev[(size_t)d].stream, hipStreamNonBlocking) != hipSuccess) { f3ds_multi_destroy(m); return F3DS_ERR_HIP; }
    }
    // development / single-GPU test boxes: F3DS_MULTI_FORCE_RCCL=1 builds the communicator with one device too and sends the
    // label block through RCCL to itself, so that the library loading and the grouped send/recv are exercised on one GPU
    if (!logical && (n_devices > 1 || f3ds::dev_getenv("F3DS_MULTI_FORCE_RCCL"))) {
        if (!g_rccl.load()) { g_multi_error = "librccl not found (F3DS_RCCL_LIB, librccl.so.1, /opt/rocm/lib)"; f3ds_multi_destroy(m); return F3DS_ERR_UNSUPPORTED; }
        std::vector<int> ids;
        for (const PerDevice& p : m->dev) ids.push_back(p.device);
        m->comm.assign((size_t)n_devices, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), n_devices, ids.data());
        if (r != ncclSuccess) { g_multi_error = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r); m->comm.clear(); f3ds_multi_destroy(m); return F3DS_ERR_HIP; }
    }
    for (int d = 0; d < n_devices; ++d) m->dev[(size_t)d].worker = std::thread(worker_loop, m, d);
    m->exchanger = std::thread(exchanger_loop, m);
    *out = m;
    return F3DS_OK;
}

// label blocks (both slots of every device) for batches of max_frames_per_device frames of up to max_points_per_frame points each,
// allocated now instead of inside the first batches
int f3ds_multi_reserve(f3ds_multi* m, size_t max_points_per_frame) {
    if (!m || !max_points_per_frame) return F3DS_ERR_ARG;
    DeviceGuard guard;
    std::unique_lock<std::mutex> lk(m->mu);
    for (auto& j : m->ring) if (j && !j->done) return F3DS_ERR_BUSY;
    const size_t per_dev = (size_t)m->max_frames_per_device * max_points_per_frame;
    for (size_t d = 0; d < m->dev.size(); ++d) {
        PerDevice& p = m->dev[d];
        if (hipSetDevice(p.device) != hipSuccess) return F3DS_ERR_HIP;
        for (int s = 0; s < SLOTS; ++s)
            if (grow_block(&p.labels[s], &p.labels_cap[s], d == 0 ? per_dev * m->dev.size() : per_dev)) { g_multi_error = "hipMalloc(label block)"; return F3DS_ERR_HIP; }
        if (d == 0 && m->dev.size() == 1 && !m->comm.empty() && grow_block(&p.loop, &p.loop_cap, per_dev)) return F3DS_ERR_HIP;
    }
    return F3DS_OK;
}

int f3ds_multi_devices(const f3ds_multi* m) { return m ? (int)m->dev.size() : 0; }

// device index (position in the devices array) a frame runs on, and its position among that device's frames
int f3ds_multi_device_of_frame(const f3ds_multi* m, int frame) { return m && frame >= 0 ? frame % (int)m->dev.size() : -1; }

int f3ds_multi_submit(f3ds_multi* m, const void* const* points, const size_t* counts, int n_frames, const f3ds_params* params,
                      uint32_t* const* point_labels, f3ds_result* results, int* ticket) {
    if (!m || !points || !counts || !params || n_frames < 0 || !ticket) return F3DS_ERR_ARG;
    *ticket = -1;
    const int G = (int)m->dev.size();
    if (n_frames > G * m->max_frames_per_device) return F3DS_ERR_CAPACITY;
    for (int i = 0; i < n_frames; ++i) if ((!points[i] && counts[i]) || counts[i] > 0x7fffffffull) return F3DS_ERR_ARG;
    std::unique_lock<std::mutex> lk(m->mu);
    if (m->broken.load()) { g_multi_error = "the RCCL communicators of this driver were aborted after an error"; return F3DS_ERR_HIP; }
    const int t = m->next_ticket, slot = t % SLOTS;
    if (m->ring[slot] && !m->ring[slot]->collected) return F3DS_ERR_BUSY;      // two batches in flight: collect the older one first (its status lives in this slot)
    std::unique_ptr<Job> j(new Job);
    j->ticket = t; j->slot = slot; j->n_frames = n_frames; j->params = *params; j->results = results;
    j->points.assign(points, points + n_frames); j->counts.assign(counts, counts + n_frames);
    j->labels.assign((size_t)n_frames, nullptr);
    if (point_labels) for (int i = 0; i < n_frames; ++i) j->labels[(size_t)i] = point_labels[i];
    j->mine.assign((size_t)G, {});
    for (int i = 0; i < n_frames; ++i) j->mine[(size_t)(i % G)].push_back(i);
    j->block.assign((size_t)G, 0); j->base.assign((size_t)G, 0); j->off.assign((size_t)n_frames, 0);
    for (int d = 0; d < G; ++d) for (int i : j->mine[(size_t)d]) { j->off[(size_t)i] = j->block[(size_t)d]; j->block[(size_t)d] += counts[i]; }
    for (int d = 0; d < G; ++d) { j->base[(size_t)d] = j->total; j->total += j->block[(size_t)d]; }       // device 0's gathered block: [dev 0 | dev 1 | ...]
    j->devices_left.store(G);
    Job* raw = j.get();
    m->ring[slot] = std::move(j);
    m->next_ticket++;
    for (int d = 0; d < G; ++d) m->dev[(size_t)d].queue.push_back(raw);
    lk.unlock();
    m->cv_work.notify_all();
    *ticket = t;
    return F3DS_OK;
}

int f3ds_multi_collect(f3ds_multi* m, int ticket) {
    if (!m || ticket < 0) return F3DS_ERR_ARG;
    std::unique_lock<std::mutex> lk(m->mu);
    Job* j = m->ring[ticket % SLOTS].get();
    if (!j || j->ticket != ticket || j->collected) return F3DS_ERR_ARG;
    m->cv_done.wait(lk, [&] { return j->done; });
    j->collected = true;
    if (j->rc) g_multi_error = j->err;
    return j->rc;
}

int f3ds_multi_segment(f3ds_multi* m, const void* const* points, const size_t* counts, int n_frames, const f3ds_params* params,
                       uint32_t* const* point_labels, f3ds_result* results) {
    if (!m || !points || !counts || !params || n_frames < 0) return F3DS_ERR_ARG;
    if (n_frames > (int)m->dev.size() * m->max_frames_per_device) return F3DS_ERR_CAPACITY;
    if (n_frames == 0) return F3DS_OK;
    int ticket = -1;
    const int rc = f3ds_multi_submit(m, points, counts, n_frames, params, point_labels, results, &ticket);
    if (rc) return rc;
    return f3ds_multi_collect(m, ticket);
}

// the gathered label block on device 0 of the batch gathered last (device pointer, uint32 per point; frames of device d at
// their running offsets inside [dev 0 | dev 1 | ...]) -- for callers that keep the labels on the GPU; valid until the
// batch after the next one is submitted (two slots)
const uint32_t* f3ds_multi_gathered_labels(const f3ds_multi* m) {
    if (!m || m->dev.empty()) return nullptr;
    const int s = m->last_slot.load();
    return s < 0 ? nullptr : m->dev[0].labels[s];
}
// the same for a given batch: NULL unless that batch has been gathered and its slot not handed to a later batch yet (the block is
// overwritten once the batch submitted two calls after it starts computing on devices[0])
const uint32_t* f3ds_multi_gathered_labels_of(f3ds_multi* m, int ticket) {
    if (!m || m->dev.empty() || ticket < 0) return nullptr;
    std::lock_guard<std::mutex> lk(m->mu);
    const Job* j = m->ring[ticket % SLOTS].get();
    if (!j || j->ticket != ticket || !j->done || j->rc) return nullptr;
    return m->dev[0].labels[j->slot];
}

}  // extern "C"
