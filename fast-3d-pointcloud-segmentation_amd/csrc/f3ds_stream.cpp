// Frame pipeline over the public entry points (include/f3ds.h, "frame pipeline"): a ring of slots, each with a
// context and pinned staging, and a few host threads that turn whatever is queued into f3ds_segment_batch calls.
// Row N4 of SURVEY.md section 8: the reference itself processes one frame per main() iteration
// (src/supervoxel_clustering.cpp:303-469) and points at a ROS node for streams (README.md:72).
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/f3ds.h"
#include "f3ds_dev.h"

namespace {
enum { FREE = 0, QUEUED, RUNNING, DONE };
struct Slot {
    f3ds_ctx* ctx = nullptr;
    void* h_pts = nullptr; size_t pts_cap = 0;          // pinned, grow-only
    uint32_t* h_lab = nullptr; size_t lab_cap = 0;
    size_t n = 0; f3ds_params prm; uint64_t tag = 0; f3ds_result res; int rc = 0; int state = FREE;
};
}  // namespace

struct f3ds_stream {
    int device = 0, depth = 0, max_batch = 1;
    std::vector<Slot> slots;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    uint64_t submitted = 0, started = 0, taken = 0;     // frame counters; frame f lives in slot f % depth
    int running = 0;                                    // batch calls in flight
    int linger_us = 300;                                // how long a frame may wait for company while other batches run (F3DS_STREAM_LINGER_US)
    bool stop = false;
};

namespace {
int pinned_grow(void** p, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return F3DS_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = bytes + bytes / 4;
    if (hipHostMalloc(p, want, hipHostMallocDefault) != hipSuccess) { *p = nullptr; return F3DS_ERR_HIP; }
    *cap = want;
    return F3DS_OK;
}

void worker(f3ds_stream* s) {
    (void)hipSetDevice(s->device);
    std::vector<f3ds_ctx*> ctxs; std::vector<const void*> pts; std::vector<size_t> cnt; std::vector<uint32_t*> lab; std::vector<f3ds_result> res;
    std::unique_lock<std::mutex> lk(s->m);
    for (;;) {
        s->cv_work.wait(lk, [&] { return s->stop || s->started < s->submitted; });
        if (s->stop) return;
        // Take what is queued at once when nothing else is running (a lone frame's latency), else while frames keep
        // arriving let the batch fill: a batch of one started on the first frame of a burst would leave the rest of the
        // burst waiting for the next idle thread.
        bool take = false;
        while (!s->stop && s->started < s->submitted) {
            if (s->submitted - s->started >= (uint64_t)s->max_batch || s->running == 0) { take = true; break; }
            const uint64_t seen = s->submitted;
            s->cv_work.wait_for(lk, std::chrono::microseconds(s->linger_us));
            if (s->started < s->submitted && s->submitted == seen) { take = true; break; }
        }
        if (s->stop) return;
        if (!take) continue;
        // the run of queued frames from the oldest one on, as long as their parameters agree
        const uint64_t first = s->started;
        const f3ds_params prm = s->slots[first % s->depth].prm;
        uint64_t end = first;
        while (end < s->submitted && end - first < (uint64_t)s->max_batch && !memcmp(&s->slots[end % s->depth].prm, &prm, sizeof prm)) ++end;
        ctxs.clear(); pts.clear(); cnt.clear(); lab.clear();
        for (uint64_t f = first; f < end; ++f) {
            Slot& sl = s->slots[f % s->depth];
            sl.state = RUNNING;
            ctxs.push_back(sl.ctx); pts.push_back(sl.h_pts); cnt.push_back(sl.n); lab.push_back(sl.h_lab);
        }
        s->started = end;
        ++s->running;
        lk.unlock();
        res.assign(ctxs.size(), f3ds_result{});
        int rc = f3ds_segment_batch(ctxs.data(), (int)ctxs.size(), pts.data(), cnt.data(), 0, &prm, lab.data(), 0, res.data());
        if (rc != F3DS_OK && ctxs.size() > 1) {          // which frame was it?  one by one, so that the others still get their labels
            for (size_t i = 0; i < ctxs.size(); ++i) {
                int r1 = f3ds_segment(ctxs[i], pts[i], cnt[i], 0, &prm, lab[i], 0, &res[i]);
                Slot& sl = s->slots[(first + i) % s->depth];
                sl.rc = r1;
            }
            rc = F3DS_OK;
        } else
            for (uint64_t f = first; f < end; ++f) s->slots[f % s->depth].rc = rc;
        lk.lock();
        --s->running;
        for (uint64_t f = first; f < end; ++f) { Slot& sl = s->slots[f % s->depth]; sl.res = res[f - first]; sl.state = DONE; }
        s->cv_done.notify_all();
    }
}
}  // namespace

extern "C" {

int f3ds_stream_create(int device, int depth, int groups, f3ds_stream** out) {
    if (!out) return F3DS_ERR_ARG;
    *out = nullptr;
    if (depth <= 0 || depth > 4096 || groups < 0) return F3DS_ERR_ARG;
    if (groups == 0) groups = depth < 4 ? depth : 4;      // libf3ds runs up to four batch calls on distinct hardware queues
    if (groups > depth) groups = depth;
    f3ds_stream* s = new f3ds_stream;
    s->device = device; s->depth = depth; s->max_batch = (depth + groups - 1) / groups;
    s->slots.resize(depth);
    if (const char* e = f3ds::dev_getenv("F3DS_STREAM_LINGER_US")) s->linger_us = atoi(e) > 0 ? atoi(e) : 0;
    for (Slot& sl : s->slots) {
        int rc = f3ds_create(device, &sl.ctx);
        if (rc) { f3ds_stream_destroy(s); return rc; }
    }

    for (int g = 0; g < groups; ++g) s->workers.emplace_back(worker, s);
    *out = s;
    return F3DS_OK;
}

void f3ds_stream_destroy(f3ds_stream* s) {
    if (!s) return;
    { std::lock_guard<std::mutex> lk(s->m); s->stop = true; }
    s->cv_work.notify_all();
    for (std::thread& t : s->workers) t.join();          // a worker finishes the batch it is in
    (void)hipSetDevice(s->device);
    for (Slot& sl : s->slots) {
        if (sl.ctx) f3ds_destroy(sl.ctx);
        if (sl.h_pts) (void)hipHostFree(sl.h_pts);
        if (sl.h_lab) (void)hipHostFree(sl.h_lab);
    }

    delete s;
}

int f3ds_stream_buffer(f3ds_stream* s, size_t n, void** points16) {
    if (!s || !points16 || n > 0x7fffffffull) return F3DS_ERR_ARG;
    *points16 = nullptr;
    Slot* sl;
    { std::lock_guard<std::mutex> lk(s->m); if (s->submitted - s->taken >= (uint64_t)s->depth) return F3DS_ERR_BUSY; sl = &s->slots[s->submitted % s->depth]; }
    (void)hipSetDevice(s->device);
    int rc = pinned_grow(&sl->h_pts, &sl->pts_cap, (n ? n : 1) * 16);      // the slot is FREE: no worker looks at it
    if (rc) return rc;
    *points16 = sl->h_pts;
    return F3DS_OK;
}

int f3ds_stream_submit(f3ds_stream* s, const void* points, size_t n, const f3ds_params* params, uint64_t tag) {
    if (!s || !params || (!points && n) || n > 0x7fffffffull) return F3DS_ERR_ARG;
    if (!(params->voxel_res > 0) || !(params->seed_res > 0)) return F3DS_ERR_ARG;
    void* buf;
    int rc = f3ds_stream_buffer(s, n, &buf);
    if (rc) return rc;
    Slot& sl = s->slots[s->submitted % s->depth];          // only this (producer) thread advances `submitted`
    if ((rc = pinned_grow((void**)&sl.h_lab, &sl.lab_cap, (n ? n : 1) * 4))) return rc;
    if (n && points != buf) memcpy(buf, points, n * 16);
    sl.n = n; sl.prm = *params; sl.tag = tag; sl.rc = 0;
    { std::lock_guard<std::mutex> lk(s->m); sl.state = QUEUED; ++s->submitted; }
    s->cv_work.notify_one();
    return F3DS_OK;
}

int f3ds_stream_next(f3ds_stream* s, uint32_t* point_labels, size_t cap, size_t* n_out, uint64_t* tag, f3ds_result* result, int wait) {
    if (!s) return F3DS_ERR_ARG;
    std::unique_lock<std::mutex> lk(s->m);
    if (s->taken == s->submitted) return F3DS_ERR_EMPTY;
    Slot& sl = s->slots[s->taken % s->depth];
    if (n_out) *n_out = sl.n;
    if (tag) *tag = sl.tag;
    if (sl.state != DONE) {
        if (!wait) return F3DS_ERR_BUSY;
        s->cv_done.wait(lk, [&] { return sl.state == DONE; });
    }
    if (point_labels && cap < sl.n) return F3DS_ERR_CAPACITY;
    lk.unlock();
    if (point_labels && sl.n && sl.rc == F3DS_OK) memcpy(point_labels, sl.h_lab, sl.n * 4);
    if (result) *result = sl.res;
    const int rc = sl.rc;
    lk.lock();
    sl.state = FREE; ++s->taken;
    return rc;
}

int f3ds_stream_peek(f3ds_stream* s, const uint32_t** point_labels, size_t* n_out, uint64_t* tag, f3ds_result* result, int wait) {
    if (!s) return F3DS_ERR_ARG;
    if (point_labels) *point_labels = nullptr;
    std::unique_lock<std::mutex> lk(s->m);
    if (s->taken == s->submitted) return F3DS_ERR_EMPTY;
    Slot& sl = s->slots[s->taken % s->depth];
    if (n_out) *n_out = sl.n;
    if (tag) *tag = sl.tag;
    if (sl.state != DONE) {
        if (!wait) return F3DS_ERR_BUSY;
        s->cv_done.wait(lk, [&] { return sl.state == DONE; });
    }
    if (point_labels && sl.rc == F3DS_OK) *point_labels = sl.h_lab;
    if (result) *result = sl.res;
    return sl.rc;
}

int f3ds_stream_pending(f3ds_stream* s) {
    if (!s) return F3DS_ERR_ARG;
    std::lock_guard<std::mutex> lk(s->m);
    return (int)(s->submitted - s->taken);
}

}  // extern "C"
