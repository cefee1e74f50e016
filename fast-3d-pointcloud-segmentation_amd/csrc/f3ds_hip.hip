// f3ds_hip.hip -- the MI355X (gfx950) device pipeline behind include/f3ds.h.
//
// Stage map (kernel functors d_<name> live in f3ds_kernels.inc; reference citations are in include/f3ds.h,
// csrc/f3ds_numerics.h, csrc/f3ds_algo.h):
//   0 voxelise   d_bbox -> d_grid -> d_keys -> radix sort -> d_heads/scan/d_segstart -> d_point_gather -> d_voxel_accum
//   1 neighbours d_neighbors (hash probe of the 27 cells), d_normals_t<384 | 256 threads> (two-ring ordered covariance, LDS tile)
//   2 seeds      d_chunkbox, d_seed_grow, d_seed_keys, radix sort, d_cell_hash, d_seed_nn, d_seed_filter
//   3 sweeps     per sweep: d_sweep_begin, d_sweep_R_first (pre-pass | round 0), d_sweep_R_round x2 | d_sweep_R + d_sweep_R_tail, d_sweep_claim,
//                d_centroid (both mark dirty tiles on the spot)  
//   4 summaries  d_sv_fill (payload rows + ordered leaf sums), d_edges, radix sort, d_edge_init,
//                d_edge_deltas, d_lambda / d_cdf_*, d_edge_weights
//   5 merge      d_inc_build, d_merge_il_t<4 or 8 waves, per-edge arrays in LDS or L2> (one persistent workgroup per frame), d_merge (global memory)
//   6 labels     d_relabel (union-find relabel in LDS + per-point label write; d_region_ids + d_point_labels beyond 12 k supervoxels)
// Every frame RECORDS its kernel calls; flush() zips the records of a batch into one dispatch per kernel (grid.y = frame).
//
// Layout in HBM: points stay as the caller's 16-byte records (one global_load_dwordx4 per lane);
// everything per voxel is SoA rows of 12 floats (48 B, 16-B aligned: xyz rgb normal pad) so a
// lane reads a neighbour with three dwordx4 loads; per-voxel neighbour table is V x 27 int32 (plus its 27 x V transpose).
// Float summation order is the reference's everywhere: parallel across outputs, sequential in
// reference order inside each reduction.  Built with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <mutex>
#include <set>
#include <vector>

#include "../../include/f3ds.h"
#include "f3ds_algo.h"
#include "f3ds_glasbey.h"
#include "f3ds_eval.h"
#include "f3ds_dev.h"

using namespace f3ds;

#include "f3ds_kernels.inc"

// ================================================================================================
// batched launch machinery
//
// Every kernel body above works on ONE frame.  The host records, per frame, the sequence of kernel
// calls a stage needs (functor type, grid width, LDS bytes, packed arguments) without launching
// anything; flush() then zips the per-frame sequences -- they are identical in shape, only the
// arguments differ -- into one dispatch per kernel with grid.y = number of frames.  A batch of 32
// frames therefore costs the launches of one frame, and every dispatch is 32 times wider.
// ================================================================================================
namespace {

template <class... Ts> struct ArgPack;
template <> struct ArgPack<> {};
template <class T, class... Ts> struct ArgPack<T, Ts...> { T head; ArgPack<Ts...> tail; };

template <class K, class... Done>
__device__ __forceinline__ void unpack_call(const ArgPack<>&, Done... d) { K{}(d...); }
template <class K, class T, class... Ts, class... Done>
__device__ __forceinline__ void unpack_call(const ArgPack<T, Ts...>& p, Done... d) { unpack_call<K>(p.tail, d..., p.head); }

// A kernel body may ask for a register budget (WAVES_PER_SIMD = w: at most 512 / w VGPRs) so that workgroups of OTHER kernels fit beside its own
// on a CU: the long-running, latency-bound kernels (the merge loop, the voxel normals) otherwise fence whole CUs off for the wide ones.
template <class K, class = void> struct waves_per_simd { static constexpr int v = 1; };
template <class K> struct waves_per_simd<K, std::void_t<decltype(K::WAVES_PER_SIMD)>> { static constexpr int v = K::WAVES_PER_SIMD; };
// The single-workgroup-per-frame kernels (serial scans, the seed-grid growth, the sweep's device-side decisions) sit on every call's critical path and do little:
// with other calls' wide kernels on the chip their waves are raised in priority (F3DS_EXP_NO_LATENCY_PRIO: A/B)
template <class K, class = void> struct is_latency_kernel { static constexpr bool v = false; };
template <class K> struct is_latency_kernel<K, std::void_t<decltype(K::LATENCY_KERNEL)>> { static constexpr bool v = K::LATENCY_KERNEL; };
template <class K, class Pack>
__global__ __launch_bounds__(K::BLOCK, waves_per_simd<K>::v) void k_batched(const Pack* frames) {
#ifndef F3DS_EXP_NO_LATENCY_PRIO
    if constexpr (is_latency_kernel<K>::v) __builtin_amdgcn_s_setprio(3);
#endif
    const Pack p = frames[f3ds_frame()];      // XCD-aware (frame, block) mapping: f3ds_kernels.inc
    unpack_call<K>(p);
}

template <class F> struct op_traits;
template <class C, class... Ps> struct op_traits<void (C::*)(Ps...) const> { using pack = ArgPack<std::decay_t<Ps>...>; };
template <class K> using pack_of = typename op_traits<decltype(&K::operator())>::pack;

inline void fill_pack(ArgPack<>&) {}
template <class T, class... Ts, class A, class... As>
inline void fill_pack(ArgPack<T, Ts...>& p, A a, As... as) { p.head = (T)a; fill_pack(p.tail, as...); }

typedef hipError_t (*LaunchFn)(uint32_t gx, uint32_t nf, uint32_t lds, hipStream_t st, const void* dargs);
template <class K>
hipError_t launch_fn(uint32_t gx, uint32_t nf, uint32_t lds, hipStream_t st, const void* dargs) {
    using Pack = pack_of<K>;
    if (lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_batched<K, Pack>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_batched<K, Pack>), dim3(gx ? gx : 1u, nf), dim3(K::BLOCK), lds, st, (const Pack*)dargs);
    return hipGetLastError();
}
struct Cmd { LaunchFn fn; uint32_t gx, lds, bytes, off; };

// dirty-tile sweeps: a sweep that changed at most V >> shift voxels lets the next one skip clean tiles.
// F3DS_INC_SHIFT=-1 turns the skipping off (every sweep evaluates every voxel), 32 forces it always.
const int g_inc_shift = [] { const char* e = dev_getenv("F3DS_INC_SHIFT"); return e ? atoi(e) : 6; }();

// Development / test switches (none is needed in production and none changes results).  The environment is read ONCE per entry
// call of the library (f3ds_segment_batch, f3ds_recluster, f3ds_refine_supervoxels) into this per-thread struct -- not per frame on the hot path,
// where hundreds of getenv() scans per call would also race with a setenv from another thread.  Tests still see per-call values.
struct Switches {
    bool direct_labels = false, copy_stream = true, copy_duplex = false, split_voxel_accum = false, sweep_tiles = true, merge_spec = true, force_global_merge = false, no_stream_pool = false, sort_pairs = false, host_prof = false, trace_err = false, vox_hash = true, vox_tiles_forced = false;
    int normals_threads = 0, merge_nw = 0, merge_keys = -1; uint32_t tile_holes = 0, ilist_slack = 32, r_rounds = F3DS_R_ROUNDS; long relabel_lds_cap = -1;
    void read() {
        auto on = [](const char* n) { return dev_getenv(n) != nullptr; };
        auto num = [](const char* n, long dflt) { const char* e = dev_getenv(n); return e ? atol(e) : dflt; };
        direct_labels = num("F3DS_DIRECT_LABELS", 0) != 0; copy_stream = num("F3DS_COPY_STREAM", 1) != 0; copy_duplex = num("F3DS_COPY_STREAM", 1) == 2; split_voxel_accum = on("F3DS_SPLIT_VOXEL_ACCUM"); sweep_tiles = num("F3DS_SWEEP_TILES", 1) != 0; merge_spec = num("F3DS_MERGE_SPEC", 1) != 0;
        force_global_merge = on("F3DS_FORCE_GLOBAL_MERGE"); no_stream_pool = on("F3DS_NO_STREAM_POOL"); sort_pairs = on("F3DS_SORT_PAIRS");
        host_prof = on("F3DS_HOST_PROF"); trace_err = on("F3DS_TRACE_ERR");
        vox_tiles_forced = num("F3DS_VOX_TILES", 1) == 2;      // (2: also for lone frames -- tests)
        vox_hash = num("F3DS_VOX_TILES", 1) != 0 && !sort_pairs && !split_voxel_accum;      // (0: stage 0 by sorting the points, as until round 4; the two development switches of that path imply it)
        normals_threads = (int)num("F3DS_NORMALS_THREADS", 0); tile_holes = (uint32_t)num("F3DS_SWEEP_TILE_HOLES", 0);
        { const long v = num("F3DS_MERGE_NW", 0); merge_nw = v == 4 ? 4 : (v ? 8 : 0); }
        { const char* e = dev_getenv("F3DS_MERGE_KEYS"); merge_keys = !e ? -1 : (!strcmp(e, "lds") ? 2 : (!strcmp(e, "global") ? 1 : 0)); }
        relabel_lds_cap = num("F3DS_RELABEL_LDS_CAP", -1);
        { const long v = num("F3DS_R_ROUNDS_RUN", F3DS_R_ROUNDS); r_rounds = v >= 1 && v <= F3DS_R_ROUNDS ? (uint32_t)v : (uint32_t)F3DS_R_ROUNDS; }
        { const long v = num("F3DS_ILIST_SLACK", 32); ilist_slack = v >= 1 && v <= 32 ? (uint32_t)v : 32u; }      // tests: a short incident-list pool (the merge stage then reruns with a larger one)
    }
};
thread_local Switches g_sw;
// where the per-edge arrays of a 4-wave merge loop live when the call shares the device with other batch calls: 0 = global memory (52 KB of LDS per
// workgroup: two fit a unit, and a voxel-normal workgroup beside them), 2 = LDS (up to 140 KB).  F3DS_MERGE_SHARED_RES=0|2 (A/B runs), read once.
const int g_merge_shared_res = [] { const char* e = dev_getenv("F3DS_MERGE_SHARED_RES"); return e && atoi(e) == 2 ? 2 : (e && atoi(e) == 0 ? 0 : 0); }();

}  // namespace

struct f3ds_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[11] = {};            // 0..7 stage boundaries, 9 / 10 around the d_normals launch
    DevCounters* d_dc = nullptr;
    DevCounters* h_dc = nullptr;       // pinned
    GridInfo* d_grid = nullptr;
    GridInfo* h_grid = nullptr;        // pinned (only read by f3ds_get_debug)
    SeedGrid* d_sgrid = nullptr;
    // recorded, not yet launched kernel calls of this frame
    std::vector<Cmd> cmds;
    std::vector<unsigned char> blob;
    std::vector<uintptr_t> pend;       // device addresses the recorded calls refer to (pointer arguments; every aligned word of struct arguments)
    MultiOp ops = {}; uint32_t ops_grid = 0;      // fills / copies recorded in a row and not yet turned into a d_multi_op call (flush_ops)
    // pinned + device staging for the packed arguments of a batch (owned by the batch's first context)
    unsigned char* h_args[2] = {nullptr, nullptr}; unsigned char* d_args[2] = {nullptr, nullptr}; size_t args_cap = 0;      // two arenas, used in turn: a flush never waits for the stream
    hipEvent_t ev_args[2] = {nullptr, nullptr}; bool args_used[2] = {false, false}; int args_flip = 0;
    hipEvent_t ev_copy[3] = {nullptr, nullptr, nullptr};      // uploads queued on the device's copy stream / labels ready on the call's stream / downloads done on the copy stream
    DevCounters* d_dcblk = nullptr; DevCounters* h_dcblk = nullptr; size_t dcblk_cap = 0;      // the batch's counters, one slot per frame
    // frame state
    bool have_frame = false;
    bool live = false;
    int rc = 0;
    f3ds_params prm;
    FrameArgs fa;
    const P16* d_pts = nullptr;
    uint32_t n = 0, V = 0, C = 0, S0 = 0, E = 0, hmask = 0;
    uint64_t *ks = nullptr; uint32_t *vs = nullptr;      // sorted point keys / indices
    uint64_t *cks = nullptr; uint32_t *cvs = nullptr;    // sorted seed-cell keys / voxels
    uint64_t *eks = nullptr;                             // sorted edge keys
    f3ds_result res;
    bool merge_in_lds = false;
    int merge_kind = 0;                // which merge kernel the last cluster stage ran (MergeKind)
    uint32_t ev_mult = 64;             // weight-history events per initial edge the merge loop may write
    uint32_t pool_mult = 1;            // leaf pool size factor (grown on demand like ev_mult)
    uint32_t vox_cap = 0;              // bound on the leaves of the current frame that stage 0's tile path sized its buffers by
    bool vox_hashed = false;           // stage 0 of the current frame took the tile path (seg_vox_tiles)
    bool vox_dense = false;            // this context met a frame the tile path refuses (an unorganised cloud, a voxel of more than VL_MAX_RUN points): its frames take the sort path ...
    uint32_t vox_dense_wait = 0, vox_dense_penalty = 8;      // ... for this many calls, then the tile path is tried again (8, 16, ... 1024 calls after every further refusal): one odd frame
                                                             // does not cost a long-lived context the faster stage 0 for good
    uint32_t ilist_mult = 1;           // incident-list pool size factor (its own: a leaf-pool overflow must not grow the list pool too)
    uint32_t edge_mult = 32;           // adjacency list room per seed (S0 * edge_mult + 1024), grown on demand
    bool relabel_lds = true;           // stage 6 as one kernel (the region-id table fits LDS for every frame of the batch)
    int refined_itr = -1;              // >= 0: the r_* buffers hold the state after that many refinement iterations of this frame
    int idxbits = -1;                  // >= 0: the sorted point keys carry the point index in their low bits
    uint32_t* user_labels = nullptr;   // device label buffer of the caller for the current call (else c->labels + copy)
    MergeDev mdev; MergeLds mlds; float host_lambda = 0.5f;
    // device scratch (grow-only)
    Buf pts, spts, keys0, keys1, vals0, vals1, flags, incl, tiles, hist, seg_start, pt_voxel, labels;
    Buf vkey, vcount, vf, nbr, nbrT, hkeys, hvals, boxes, ckey, cell_start, chk, chv, chvals, seed_orig, keep, seed_kept;
    Buf owner0, owner1, ownR, dist0, dist1, R, hc, hcount, hlo, hhi, ghost_vox, ghost_active, ghost_done, ghost_head, ghost_next;
    Buf loff, rows, row_voxel, racc0, rcnt0, rrec0, ralive0, ehk, ekeys0, ekeys1, evals0, evals1, ea0, eb0;
    Buf ea, eb, ew, eku, ehist, ealive, ev_epoch, ev_key, ev_prev, racc, rcnt, rrec, ralive, rhead, rtail, lnext, parent, markA, markB, tl, merges;
    Buf r_vf, r_owner, r_dist, r_hc, r_hcount, r_hlo, r_hhi, r_gvox, r_gact, r_gdone, r_ghead, r_gnext, r_tl, r_tcnt, r_seed, r_L;      // refineSupervoxels works on copies
    Buf tstamp, tround, hdirty, htiles, htcnt, vwl, vwl2, vtmask, glut, truth_pts, tsum, tcol, tlab, ctab, csize, eroot, eincl;      // ground-truth evaluation
    Buf hcnt, pslot, vlist;      // stage 0, tile path: per (tile, entry) point count / list base / leaf ordinal; per point its (entry, rank in tile); per-leaf point lists
    Buf u_src, u_voff, u_xyz, u_rgba, u_cent, u_nrm;      // f3ds_cluster_supervoxels: the caller's supervoxels as uploaded
    Buf deltas, skeys0, skeys1, svals0, svals1, cdf_hist, cdf, root, rrank, pool, rstart, rnleaf, rcap, tile_n1, tile_ord, tile_slots, ilist, istart, ilen, icap, rincl;      // (rincl stays last: f3ds_destroy walks pts..rincl)
    std::vector<uint32_t> tsize;       // voxels per truth label (evaluation)
    // f3ds_cluster_supervoxels: the state is caller-supplied supervoxels (no points, no voxel grid).  user_label[h] = the caller's label of internal
    // supervoxel h (h = rank in ascending label + 1; [0] = 0), user_row[h] = its row in the caller's arrays; empty after f3ds_segment
    bool user_mode = false;
    std::vector<uint32_t> user_label, user_row;
};

namespace {

// A recorded, not yet flushed kernel call holds raw device pointers: a buffer such a call refers to must not be freed and
// reallocated before the flush (the dispatch would write into freed memory).  rec<>() notes the address of every pointer-typed
// argument (and, for the few struct arguments -- SweepFrame, MergeDev, MergeLds --, every 8-byte word of the struct, which is
// where their pointer members sit); scalar arguments are never mistaken for addresses.  Checked on every regrow.
bool referenced_by_pending_calls(const f3ds_ctx* c, const Buf& b) {
    const uintptr_t lo = (uintptr_t)b.p, hi = lo + b.cap;
    for (const uintptr_t w : c->pend) if (w >= lo && w < hi) return true;
    for (uint32_t k = 0; k < c->ops.n; ++k) {      // (fills / copies recorded and not yet merged into a call)
        const uintptr_t d = (uintptr_t)c->ops.dst[k], sp = (uintptr_t)c->ops.src[k];
        if ((d >= lo && d < hi) || (sp >= lo && sp < hi)) return true;
    }
    return false;
}
template <class A> inline void note_arg(f3ds_ctx* c, const A& a) {
    if constexpr (std::is_pointer<A>::value) { if (a) c->pend.push_back((uintptr_t)a); }
    else if constexpr (std::is_class<A>::value) {
        static_assert(std::is_trivially_copyable<A>::value, "kernel arguments must be plain data");
        for (size_t off = 0; off + 8 <= sizeof(A); off += 8) { uintptr_t w; memcpy(&w, reinterpret_cast<const unsigned char*>(&a) + off, 8); if (w) c->pend.push_back(w); }
    }
}
// Scratch is grow-only per context, and sized by the largest frame ANY context of the device has seen: g_scratch_hwm holds,
// per buffer slot, the largest request so far.  A context that has to grow a buffer -- or meets a request below the mark
// while nothing recorded refers to the buffer -- goes straight to the mark (if that is within 4x of its own request), so that
// a pool of contexts fed with frames of varying size stops allocating after every context has been used twice (hipFree waits for the whole device: a
// regrow in steady state stalls every batch in flight).
std::atomic<size_t> g_scratch_hwm[16][160];
std::atomic<unsigned long long> g_scratch_allocs{0};      // hipMalloc calls for scratch so far (F3DS_HOST_PROF prints it per batch)
template <class T>
int ensure(f3ds_ctx* c, Buf& b, size_t count, T** out) {
    size_t bytes = count * sizeof(T);
    if (bytes < 256) bytes = 256;
    size_t target = bytes;
    const ptrdiff_t slot = &b - &c->pts;
    if (slot >= 0 && slot < 160) {
        std::atomic<size_t>& hwm = g_scratch_hwm[c->device & 15][slot];
        size_t seen = hwm.load(std::memory_order_relaxed);
        while (seen < bytes && !hwm.compare_exchange_weak(seen, bytes, std::memory_order_relaxed)) {}
        if (seen > target && seen <= 4 * bytes) target = seen;      // (a frame of another scale altogether does not size this one)
    }
    // ENSURE is an idempotent getter while the buffer covers the request: contents are only ever discarded when the request does not
    // fit (the caller is about to overwrite the buffer anyway).  The device-wide mark sizes such a regrow; buffers that are merely
    // below the mark are brought up to it by pregrow_scratch() at the start of a segment call, when no buffer holds frame state.
    if (b.cap < bytes) {
        if (b.p && (!c->cmds.empty() || c->ops.n) && referenced_by_pending_calls(c, b)) {
            fprintf(stderr, "f3ds: internal error: regrowing a buffer that a recorded kernel call refers to\n");
            return F3DS_ERR_LOGIC;
        }
        g_scratch_allocs.fetch_add(1, std::memory_order_relaxed);
        static const bool trace = dev_getenv("F3DS_TRACE_ALLOC") != nullptr;
        if (trace && b.p) fprintf(stderr, "f3ds: regrow slot %td: cap %zu, request %zu, mark %zu, pending calls %zu\n", slot, b.cap, bytes, target, c->cmds.size());
        if (b.p) { HIPCHECK(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
        size_t want = target + target / 4 + 64;
        HIPCHECK(hipMalloc(&b.p, want));
        b.cap = want;
    }
    *out = reinterpret_cast<T*>(b.p);
    return F3DS_OK;
}
// Start of a segment call: nothing is recorded and no buffer holds state of a frame that will be read again (the call starts a new
// frame), so this is the one moment a context may trade a buffer for a larger one without losing anything.  Every allocated buffer
// that is below the device-wide mark of its slot (within 4x) goes to the mark: a pool of contexts fed with frames of varying size
// stops allocating once every context has been used twice (hipFree waits for the whole device: a regrow in steady state stalls
// every batch in flight).
int pregrow_scratch(f3ds_ctx* c) {
    Buf* bufs = &c->pts;
    const size_t nb = (reinterpret_cast<char*>(&c->rincl) - reinterpret_cast<char*>(&c->pts)) / sizeof(Buf) + 1;
    for (size_t i = 0; i < nb && i < 160; ++i) {
        Buf& b = bufs[i];
        if (!b.p) continue;
        const size_t mark = g_scratch_hwm[c->device & 15][i].load(std::memory_order_relaxed);
        if (b.cap >= mark || mark > 4 * b.cap) continue;
        g_scratch_allocs.fetch_add(1, std::memory_order_relaxed);
        HIPCHECK(hipFree(b.p)); b.p = nullptr; b.cap = 0;
        const size_t want = mark + mark / 4 + 64;
        HIPCHECK(hipMalloc(&b.p, want));
        b.cap = want;
    }
    return F3DS_OK;
}
#define ENSURE(buf, T, count, ptr) do { int rc_ = ensure<T>(c, buf, (size_t)(count), &ptr); if (rc_) return rc_; } while (0)

// Workgroups per frame of a wide kernel.  A launch covers all frames of the batch (grid.y = frame), so a frame gets its
// share of a launch-wide budget of ~3 k workgroups (12 per CU) and the kernels loop (grid-stride) over the rest: with
// 192 frames per launch the streaming kernels of the sweeps run 1.5-2x faster on 32 fat workgroups per frame than on
// 300 thin ones (per-workgroup prologue: argument pack, counters, stamps), see DESIGN.md 4b.  The hash-probing /
// gathering kernels want every wave they can get and keep the old cap (grid_wide).  Set per batch call (one host thread).
thread_local size_t g_grid_cap = 2048;
thread_local int g_batch_frames = 1;      // frames of the batch call this thread is running
// batch calls inside f3ds_segment_batch right now, per device: a call whose merge dispatch shares the chip with other calls takes the 4-wave merge kernel (choose_merge_kind)
std::atomic<int> g_batch_calls[16];
struct BatchCallCount { int d; explicit BatchCallCount(int dev) : d(dev & 15) { g_batch_calls[d].fetch_add(1, std::memory_order_relaxed); } ~BatchCallCount() { g_batch_calls[d].fetch_sub(1, std::memory_order_relaxed); } };
size_t grid_cap_for_batch(int frames) {
    static const size_t target = dev_getenv("F3DS_GRID_TARGET") ? (size_t)atol(dev_getenv("F3DS_GRID_TARGET")) : 3072;      // with six calls in flight: 24 576: 2 150, 12 288: 2 250, 6 144: 2 280, 3 072 ... 1 024: 2 340 Mpoints/s
    if (!target) return 2048;
    size_t cap = target / (size_t)(frames > 0 ? frames : 1);
    return cap < 8 ? 8 : (cap > 2048 ? 2048 : cap);
}
inline uint32_t grid_wide(size_t work, int block) {
    size_t g = (work + block - 1) / block;
    return (uint32_t)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
inline uint32_t grid_for(size_t work, int block) {
    size_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > g_grid_cap) g = g_grid_cap;
    return (uint32_t)g;
}
inline uint32_t pow2_ge(size_t x) { uint32_t p = 1; while (p < x) p <<= 1; return p; }
int bits_for(uint64_t max_value) { int b = 0; while (b < 64 && (max_value >> b)) ++b; return b; }

// record one kernel call of frame `c` (nothing is launched here)
template <class K, class... As> void rec(f3ds_ctx* c, uint32_t gx, uint32_t lds, As... as);
// the fills / copies recorded since the last other call become one d_multi_op call (every frame of a batch records the same sequence, so the same grouping)
inline void flush_ops(f3ds_ctx* c) {
    if (!c->ops.n) return;
    const MultiOp m = c->ops; const uint32_t g = c->ops_grid;
    c->ops.n = 0; c->ops_grid = 0;
    rec<d_multi_op>(c, g, 0u, m);
}
template <class K, class... As>
void rec(f3ds_ctx* c, uint32_t gx, uint32_t lds, As... as) {
    if constexpr (!std::is_same<K, d_multi_op>::value) flush_ops(c);
    using Pack = pack_of<K>;
    static_assert(std::is_trivially_copyable<Pack>::value, "kernel arguments must be plain data");
    Pack p;
    memset(&p, 0, sizeof p);
    fill_pack(p, as...);
    (note_arg(c, as), ...);
    Cmd cmd; cmd.fn = &launch_fn<K>; cmd.gx = gx; cmd.lds = lds; cmd.bytes = (uint32_t)sizeof(Pack); cmd.off = (uint32_t)c->blob.size();
    c->blob.resize(c->blob.size() + sizeof(Pack));
    memcpy(c->blob.data() + cmd.off, &p, sizeof(Pack));
    c->cmds.push_back(cmd);
}
inline void rec_op(f3ds_ctx* c, void* dst, const void* src, uint32_t value, size_t bytes) {
    const uint32_t words = (uint32_t)((bytes + 3) / 4);
    if (c->ops.n == (uint32_t)MULTI_OPS) flush_ops(c);
    MultiOp& m = c->ops;
    if (m.n == 0) memset(&m, 0, sizeof m);
    m.dst[m.n] = (uint32_t*)dst; m.src[m.n] = (const uint32_t*)src; m.val[m.n] = value; m.words[m.n] = words; m.n++;
    const uint32_t g = grid_for(words, 256);
    if (g > c->ops_grid) c->ops_grid = g;
}
inline void rec_fill(f3ds_ctx* c, void* p, uint32_t value, size_t bytes) { rec_op(c, p, nullptr, value, bytes); }
inline void rec_copy(f3ds_ctx* c, void* dst, const void* src, size_t bytes) { rec_op(c, dst, src, 0u, bytes); }

// a batch: the frames that are still being processed together, one stream, one argument arena
struct Batch {
    std::vector<f3ds_ctx*> fr;
    hipStream_t st = nullptr;
    f3ds_ctx* owner = nullptr;     // holds the argument arena and the stage events
};

thread_local double g_t_wait = 0, g_t_launch = 0;
// F3DS_TRACE_ERR=1: say which stage refused a frame (development aid)
static int trace_err(int code, const char* where, const f3ds_ctx* c) {
    if (code && g_sw.trace_err)
        fprintf(stderr, "f3ds: error %d after %s (V %u, seeds %u, edges %u, chain overflow %d, capacity flag %d)\n", code, where, c->V, c->S0, c->h_dc->n_edges,
                c->h_dc->r_overflow, c->h_dc->ev_overflow);
    return code;
}
static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// zip the recorded calls of all live frames into batched dispatches
int flush(Batch& b, hipEvent_t before_launch = nullptr) {      // before_launch: recorded after the argument upload, right before the first dispatch
    if (b.fr.empty()) return F3DS_OK;
    for (f3ds_ctx* c : b.fr) flush_ops(c);
    const size_t ncmd = b.fr[0]->cmds.size();
    for (f3ds_ctx* c : b.fr) if (c->cmds.size() != ncmd) return F3DS_ERR_UNSUPPORTED;   // frames must take the same path
    if (ncmd == 0) return F3DS_OK;
    const uint32_t nf = (uint32_t)b.fr.size();
    size_t total = 0;
    std::vector<size_t> off(ncmd);
    for (size_t j = 0; j < ncmd; ++j) { off[j] = total; total += (((size_t)b.fr[0]->cmds[j].bytes * nf) + 63u) & ~(size_t)63u; }
    f3ds_ctx* o = b.owner;
    if (o->args_cap < total) {
        HIPCHECK(hipStreamSynchronize(b.st));
        for (int k = 0; k < 2; ++k) {
            if (o->h_args[k]) HIPCHECK(hipHostFree(o->h_args[k]));
            if (o->d_args[k]) HIPCHECK(hipFree(o->d_args[k]));
            o->h_args[k] = o->d_args[k] = nullptr; o->args_used[k] = false;
        }
        o->args_cap = total * 2;
        for (int k = 0; k < 2; ++k) {
            HIPCHECK(hipHostMalloc((void**)&o->h_args[k], o->args_cap, hipHostMallocDefault));
            HIPCHECK(hipMalloc((void**)&o->d_args[k], o->args_cap));
            if (!o->ev_args[k]) HIPCHECK(hipEventCreateWithFlags(&o->ev_args[k], hipEventDisableTiming));
        }
    }
    // The argument blocks of a flush travel pinned arena -> device arena -> kernels.  Two arena pairs are used in turn and an event marks the
    // end of the last dispatch that reads a pair, so the host packs and launches stage k+1 while stage k still runs: the only wait here is
    // for the flush before the previous one, which is long over (round 2 synchronised the stream at every flush).
    const int fl = o->args_flip; o->args_flip ^= 1;
    if (o->args_used[fl]) { const double t0 = now_ms(); HIPCHECK(hipEventSynchronize(o->ev_args[fl])); g_t_wait += now_ms() - t0; }
    unsigned char* const h_args = o->h_args[fl]; unsigned char* const d_args = o->d_args[fl];
    const double tl0 = now_ms();
    for (size_t j = 0; j < ncmd; ++j)
        for (uint32_t i = 0; i < nf; ++i) {
            const Cmd& cm = b.fr[i]->cmds[j];
            if (cm.fn != b.fr[0]->cmds[j].fn) return F3DS_ERR_UNSUPPORTED;
            memcpy(h_args + off[j] + (size_t)i * cm.bytes, b.fr[i]->blob.data() + cm.off, cm.bytes);
        }
    HIPCHECK(hipMemcpyAsync(d_args, h_args, total, hipMemcpyHostToDevice, b.st));
    if (before_launch) HIPCHECK(hipEventRecord(before_launch, b.st));
    for (size_t j = 0; j < ncmd; ++j) {
        uint32_t gx = 1, lds = 0;
        for (uint32_t i = 0; i < nf; ++i) { const Cmd& cm = b.fr[i]->cmds[j]; if (cm.gx > gx) gx = cm.gx; if (cm.lds > lds) lds = cm.lds; }
        HIPCHECK(b.fr[0]->cmds[j].fn(gx, nf, lds, b.st, d_args + off[j]));
    }
    HIPCHECK(hipEventRecord(o->ev_args[fl], b.st)); o->args_used[fl] = true;
    for (f3ds_ctx* c : b.fr) { c->cmds.clear(); c->blob.clear(); c->pend.clear(); c->ops.n = 0; c->ops_grid = 0; }
    g_t_launch += now_ms() - tl0;
    return F3DS_OK;
}
// flush, then bring every live frame's counters to the host
int flush_sync(Batch& b) {
    int rc = flush(b);
    if (rc) return rc;
    const size_t nf = b.fr.size();
    if (nf == 1) HIPCHECK(hipMemcpyAsync(b.fr[0]->h_dc, b.fr[0]->d_dc, sizeof(DevCounters), hipMemcpyDeviceToHost, b.st));
    else if (nf > 1) {
        // one gather kernel + one copy instead of a copy per frame
        f3ds_ctx* o = b.owner;
        if (o->dcblk_cap < nf) {
            HIPCHECK(hipStreamSynchronize(b.st));
            if (o->d_dcblk) HIPCHECK(hipFree(o->d_dcblk));
            if (o->h_dcblk) HIPCHECK(hipHostFree(o->h_dcblk));
            o->dcblk_cap = nf * 2;
            HIPCHECK(hipMalloc((void**)&o->d_dcblk, o->dcblk_cap * sizeof(DevCounters)));
            HIPCHECK(hipHostMalloc((void**)&o->h_dcblk, o->dcblk_cap * sizeof(DevCounters), hipHostMallocDefault));
        }
        for (size_t i = 0; i < nf; ++i) rec<d_dc_gather>(b.fr[i], 1u, 0u, (const DevCounters*)b.fr[i]->d_dc, o->d_dcblk + i);
        if ((rc = flush(b))) return rc;
        HIPCHECK(hipMemcpyAsync(o->h_dcblk, o->d_dcblk, nf * sizeof(DevCounters), hipMemcpyDeviceToHost, b.st));
    }
    { const double t0 = now_ms(); HIPCHECK(hipStreamSynchronize(b.st)); g_t_wait += now_ms() - t0; }
    HIPCHECK(hipGetLastError());
    if (nf > 1) for (size_t i = 0; i < nf; ++i) *b.fr[i]->h_dc = b.owner->h_dcblk[i];
    return F3DS_OK;
}

// inclusive scan of n uint32 values (in -> out): always the same three calls so that frames of
// different sizes record identical sequences
int scan_u32(f3ds_ctx* c, const uint32_t* in, uint32_t* out, uint32_t n) {
    const uint32_t nt = n ? (n + SCAN_TILE - 1) / SCAN_TILE : 1u;
    uint32_t* tiles;
    ENSURE(c->tiles, uint32_t, nt, tiles);
    rec<d_scan_tiles>(c, nt, 0u, in, out, tiles, n);
    rec<d_scan_single>(c, 1u, 0u, tiles, nt);
    rec<d_scan_add>(c, nt, 0u, out, (const uint32_t*)tiles, n);
    return F3DS_OK;
}
// stable sort of (key,val) pairs on the low `total_bits` bits (the same for every frame of a batch)
// (n_dev: the element count lives on the device and `n` only bounds it -- pair sorts only)
int radix_sort(f3ds_ctx* c, uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, uint32_t n, int total_bits, uint64_t** keys_out, uint32_t** vals_out, int base_shift = 0, const uint32_t* n_dev = nullptr) {
    *keys_out = k0; if (vals_out) *vals_out = v0;
    if (total_bits <= 0) return F3DS_OK;
    const int passes = (total_bits + RS_MAXBITS - 1) / RS_MAXBITS;
    const int per = (total_bits + passes - 1) / passes;
    const uint32_t nb = n ? (n + RS_TILE - 1) / RS_TILE : 1u;
    uint32_t* hist;
    ENSURE(c->hist, uint32_t, (size_t)RS_BINS * nb, hist);
    int shift = 0;
    for (int p = 0; p < passes; ++p) {
        const int bits = (total_bits - shift) < per ? (total_bits - shift) : per;
        rec<d_radix_hist>(c, nb, 0u, (const uint64_t*)k0, n, base_shift + shift, bits, hist, nb, n_dev);
        rec<d_scan_single>(c, 1u, 0u, hist, (uint32_t)((1u << bits) * nb));
        if (v0) rec<d_radix_scatter>(c, nb, 0u, (const uint64_t*)k0, (const uint32_t*)v0, k1, v1, n, base_shift + shift, bits, (const uint32_t*)hist, nb, n_dev);
        else rec<d_radix_scatter_k>(c, nb, 0u, (const uint64_t*)k0, k1, n, base_shift + shift, bits, (const uint32_t*)hist, nb);      // payload in the key's low bits
        std::swap(k0, k1); std::swap(v0, v1);
        shift += bits;
    }
    *keys_out = k0; if (vals_out) *vals_out = v0;
    return F3DS_OK;
}

// ---------------------------------------------------------------- per-frame stage recorders ------
// stage 0a: bounding box and grid
int seg_bbox(f3ds_ctx* c) {
    rec<d_init_counters>(c, 1u, 0u, c->d_dc);
    rec<d_bbox>(c, grid_for(c->n, 256 * 8) < 512 ? grid_for(c->n, 256 * 8) : 512u, 0u, c->d_pts, c->n, c->fa, c->d_dc);
    rec<d_grid>(c, 1u, 0u, c->d_dc, c->prm.voxel_res, c->d_grid);
    return F3DS_OK;
}
// stage 0b: Morton keys, stable sort, voxel segments
int seg_sort(f3ds_ctx* c, int sort_bits, int idxbits) {      // idxbits >= 0: (code << idxbits) | index in one array; -1: (key, index) pairs
    const uint32_t n = c->n;
    uint64_t *k0, *k1; uint32_t *v0 = nullptr, *v1 = nullptr, *flags, *incl, *seg_start; int* pt_voxel;
    ENSURE(c->keys0, uint64_t, n, k0); ENSURE(c->keys1, uint64_t, n, k1);
    if (idxbits < 0) { ENSURE(c->vals0, uint32_t, n, v0); ENSURE(c->vals1, uint32_t, n, v1); }
    ENSURE(c->flags, uint32_t, n, flags); ENSURE(c->incl, uint32_t, n, incl); ENSURE(c->seg_start, uint32_t, (size_t)n + 1, seg_start);
    ENSURE(c->pt_voxel, int, n, pt_voxel);
    c->idxbits = idxbits;
    const int ks = idxbits < 0 ? 0 : idxbits;
    rec<d_keys>(c, grid_for(n, 256), 0u, c->d_pts, n, c->fa, (const GridInfo*)c->d_grid, k0, v0, ks);
    c->vs = nullptr;
    int rc = radix_sort(c, k0, v0, k1, v1, n, sort_bits, &c->ks, &c->vs, ks);
    if (rc) return rc;
    const uint64_t invalid = 1ull << (3 * c->h_dc->depth);
    if (g_sw.split_voxel_accum) {      // development: round 2's chain (d_point_gather reads the rank array)
        rec<d_heads>(c, grid_for(n, 256), 0u, (const uint64_t*)c->ks, n, invalid, flags, ks);
        if ((rc = scan_u32(c, flags, incl, n))) return rc;
        rec<d_segstart>(c, grid_for(n, 256), 0u, (const uint64_t*)c->ks, (const uint32_t*)flags, (const uint32_t*)incl, n, invalid, seg_start, &c->d_dc->n_voxels, &c->d_dc->n_valid, ks);
        return F3DS_OK;
    }
    const uint32_t nt = n ? (n + SCAN_TILE - 1) / SCAN_TILE : 1u;
    uint32_t* tiles; ENSURE(c->tiles, uint32_t, nt, tiles);
    rec<d_seg_count>(c, nt, 0u, (const uint64_t*)c->ks, n, invalid, ks, tiles);
    rec<d_scan_single>(c, 1u, 0u, tiles, nt);
    rec<d_seg_write>(c, nt, 0u, (const uint64_t*)c->ks, n, invalid, ks, (const uint32_t*)tiles, seg_start, &c->d_dc->n_voxels, &c->d_dc->n_valid);
    return F3DS_OK;
}
// stage 0b + 0c without sorting the points (f3ds_kernels.inc, "stage 0 without sorting the points"): per-tile LDS grouping, sort of the tiles' descriptors, per-leaf
// point lists, ordered leaf sums.  Everything is recorded before V is known: buffers and grids are sized by bounds, the kernels read the counts from the device.
int seg_vox_tiles(f3ds_ctx* c, int code_bits, int tile_bits) {      // (code_bits, tile_bits: the same for every frame of a batch -- the frames record the same sort passes)
    const uint32_t n = c->n;
    const uint32_t ntiles = (n + VT_TILE - 1u) / VT_TILE;
    const uint64_t dcap64 = (uint64_t)ntiles * VT_ENT_MAX;
    if (code_bits + tile_bits + VT_CNT_BITS > 64 || dcap64 > 0x7fffffffull || (uint64_t)ntiles * VT_TAB > 0x7fffffffull) return F3DS_ERR_UNSUPPORTED;      // (the caller takes the sort path)
    const uint32_t dcap = (uint32_t)dcap64;
    const uint64_t fin = c->h_dc->n_finite;
    const uint32_t capV = (uint32_t)std::max<uint64_t>(256u, std::min<uint64_t>(fin, dcap));
    c->vox_cap = capV;
    uint64_t *k0, *k1; uint32_t *ploc, *v0, *v1, *part, *vkey, *vcount, *seg_start, *list; uint2* bo_te; float* vf; int* pt_voxel;
    const uint32_t nchunks = (dcap + DS_CHUNK - 1u) / DS_CHUNK;
    ENSURE(c->pslot, uint32_t, n, ploc); ENSURE(c->keys0, uint64_t, dcap, k0); ENSURE(c->keys1, uint64_t, dcap, k1); ENSURE(c->vals0, uint32_t, dcap, v0); ENSURE(c->vals1, uint32_t, dcap, v1);
    { uint32_t* w; ENSURE(c->hcnt, uint32_t, (size_t)ntiles * VT_TAB * 2 + 2u * nchunks, w); bo_te = reinterpret_cast<uint2*>(w); part = w + (size_t)ntiles * VT_TAB * 2; }
    ENSURE(c->vkey, uint32_t, (size_t)capV * 3, vkey); ENSURE(c->vcount, uint32_t, capV, vcount); ENSURE(c->vf, float, (size_t)capV * 12, vf);
    ENSURE(c->seg_start, uint32_t, (size_t)capV + 1, seg_start); ENSURE(c->vlist, uint32_t, n, list); ENSURE(c->pt_voxel, int, n, pt_voxel);
    { uint32_t *f, *incl; ENSURE(c->flags, uint32_t, capV, f); ENSURE(c->incl, uint32_t, capV, incl); }      // (the seed stage scans V / C flags through them without asking)
    c->idxbits = -1; c->ks = nullptr; c->vs = nullptr;
    rec<d_tile_keys>(c, std::min<uint32_t>(ntiles, (uint32_t)g_grid_cap * 16u), 0u, c->d_pts, n, c->fa, (const GridInfo*)c->d_grid, ploc, k0, v0, dcap, tile_bits, c->d_dc);
    uint64_t* ks; uint32_t* vs;
    int rc = radix_sort(c, k0, v0, k1, v1, dcap, code_bits + tile_bits, &ks, &vs, VT_CNT_BITS, &c->d_dc->seg_count);      // (the bits above the count field)
    if (rc) return rc;
    rec<d_desc_part>(c, nchunks, 0u, (const uint64_t*)ks, dcap, tile_bits, part, (const DevCounters*)c->d_dc);
    rec<d_desc_offsets>(c, 1u, 0u, part, dcap, capV, seg_start, c->d_dc);
    rec<d_desc_apply>(c, nchunks, 0u, (const uint64_t*)ks, (const uint32_t*)vs, dcap, capV, tile_bits, c->fa, (const GridInfo*)c->d_grid, (const uint32_t*)part, bo_te,
                      seg_start, vkey, (const DevCounters*)c->d_dc);
    rec<d_tile_place>(c, std::min<uint32_t>(ntiles, (uint32_t)g_grid_cap * 16u), 0u, (const uint32_t*)ploc, n, (const uint2*)bo_te, list, pt_voxel, (const DevCounters*)c->d_dc);
    rec<d_voxel_list_accum>(c, std::min<uint32_t>(grid_wide(capV, 256), std::max<uint32_t>(64u, grid_wide(n / 8u, 256))), 0u, c->d_pts, list, (const uint32_t*)seg_start, c->fa, vf, vcount, c->d_dc, capV);
    c->vox_hashed = true;
    return F3DS_OK;
}
// stage 0c + 1 + 2a: voxel sums, neighbour tables, normals, seed grid growth
int seg_voxels(f3ds_ctx* c) {
    const uint32_t V = c->V, n = c->n;
    uint32_t *vkey, *vcount, *hvals; float *vf; int *nbr, *nbrT; uint64_t* hkeys;
    if (c->vox_hashed) {      // the leaf sums exist already (seg_vox_tiles): the leaf keys go into the hash table, then the neighbour search
        const uint32_t hc2 = pow2_ge((size_t)V * 2 + 16);
        c->hmask = hc2 - 1;
        ENSURE(c->nbr, int, (size_t)V * 27, nbr); ENSURE(c->nbrT, int, (size_t)V * 27, nbrT); ENSURE(c->hkeys, uint64_t, hc2, hkeys); ENSURE(c->hvals, uint32_t, hc2, hvals);
        rec_fill(c, hkeys, 0xFFFFFFFFu, (size_t)hc2 * 8);
        rec<d_vox_hash>(c, 1u, 0u, (const uint32_t*)c->vkey.p, V, (const GridInfo*)c->d_grid, hkeys, hvals, c->hmask);
        rec<d_neighbors>(c, grid_wide((size_t)V * 27, 256), 0u, (const uint32_t*)c->vkey.p, (const DevCounters*)c->d_dc, (const GridInfo*)c->d_grid, (const uint64_t*)hkeys, (const uint32_t*)hvals,
                         c->hmask, nbr, nbrT);
        return F3DS_OK;
    }
    const uint32_t hcap = pow2_ge((size_t)V * 2 + 16);
    c->hmask = hcap - 1;
    ENSURE(c->vkey, uint32_t, (size_t)V * 3, vkey); ENSURE(c->vcount, uint32_t, V, vcount); ENSURE(c->vf, float, (size_t)V * 12, vf);
    ENSURE(c->nbr, int, (size_t)V * 27, nbr); ENSURE(c->nbrT, int, (size_t)V * 27, nbrT); ENSURE(c->hkeys, uint64_t, hcap, hkeys); ENSURE(c->hvals, uint32_t, hcap, hvals);
    rec_fill(c, hkeys, 0xFFFFFFFFu, (size_t)hcap * 8);
    if (g_sw.split_voxel_accum) {      // development: round 2's two kernels (a sorted copy of the frame in between)
        P16* spts; ENSURE(c->spts, P16, n, spts);
        rec<d_point_gather>(c, grid_wide(n, 256), 0u, c->d_pts, (const uint32_t*)c->vs, (const uint64_t*)c->ks, c->idxbits < 0 ? 0 : c->idxbits, (const uint32_t*)c->incl.p, n, (const DevCounters*)c->d_dc, spts, (int*)c->pt_voxel.p);
        rec<d_voxel_accum>(c, grid_wide(V, 256), 0u, (const P16*)spts, (const uint64_t*)c->ks, c->idxbits < 0 ? 0 : c->idxbits, (const uint32_t*)c->seg_start.p, (const DevCounters*)c->d_dc, c->fa,
                           (const GridInfo*)c->d_grid, vkey, vcount, vf, hkeys, hvals, c->hmask);
    } else
        rec<d_voxel_gather_accum>(c, grid_wide(V, 256), 0u, c->d_pts, (const uint32_t*)c->vs, (const uint64_t*)c->ks, c->idxbits < 0 ? 0 : c->idxbits, (const uint32_t*)c->seg_start.p, n,
                                  (const DevCounters*)c->d_dc, c->fa, (const GridInfo*)c->d_grid, vkey, vcount, vf, hkeys, hvals, c->hmask, (int*)c->pt_voxel.p);
    rec<d_neighbors>(c, grid_wide((size_t)V * 27, 256), 0u, (const uint32_t*)vkey, (const DevCounters*)c->d_dc, (const GridInfo*)c->d_grid, (const uint64_t*)hkeys, (const uint32_t*)hvals,
                     c->hmask, nbr, nbrT);
    return F3DS_OK;
}
// stage 1b: voxel normals -- ONE launch per batch call, bracketed by its own pair of events (f3ds_result.ms_stage[7]: besides the merge
// loop the only kernel of the path whose launch duration is measured live, bench.py's roofline picks the longer of the two)
int seg_normals(f3ds_ctx* c) {
    const uint32_t nt = (c->V + NT_TILE - 1) / NT_TILE;
    uint32_t *tn1, *tord, *tslots;      // the tiles' one-ring tables, built on the way for the sweeps (d_sweep_R_pre, d_sweep_claim)
    ENSURE(c->tile_n1, uint32_t, nt, tn1); ENSURE(c->tile_ord, uint32_t, (size_t)nt * NT_RING1, tord); ENSURE(c->tile_slots, uint32_t, (size_t)nt * SW_SLOT_WORDS * NT_TILE, tslots);
    // One workgroup per tile by default.  F3DS_NORMALS_WGS=<n> (experiment): a launch-wide budget of n workgroups, each walking a run of
    // tiles -- placed once, a 72 KB six-wave workgroup then keeps its compute unit instead of competing for one per tile.  With six calls in flight the
    // launch drops from 50-100 ms to 30 ms and the sweeps of the other calls grow by as much (2 140-2 155 vs 2 170 Mpoints/s on one box); alone it is
    // twice as slow (83 vs 43 us per frame: two rounds of long-lived workgroups).  The chip's compute-unit time is conserved; only work removed counts.
#ifdef F3DS_NORMALS_LOOP
    static const uint32_t budget = dev_getenv("F3DS_NORMALS_WGS") ? (uint32_t)atoi(dev_getenv("F3DS_NORMALS_WGS")) : 0u;
#else
    static const uint32_t budget = 0u;      // (the tile loop is compiled in with make EXTRA=-DF3DS_NORMALS_LOOP only)
#endif
    const uint32_t share = budget ? std::max(2u, budget / (uint32_t)g_batch_frames) : nt;
    // 256 threads per tile for the calls of a batch pipeline (their workgroups fit beside the 4-wave merge loops of the other calls), 384 otherwise (kernels.inc)
    const int nthr_env = g_sw.normals_threads;
    const uint32_t holes = g_sw.tile_holes;
    if (nthr_env ? nthr_env == 256 : g_batch_frames >= 16)
        rec<d_normals_t<256>>(c, std::min(nt, share), 0u, (float*)c->vf.p, (const int*)c->nbr.p, (const DevCounters*)c->d_dc, tn1, tord, tslots, holes);
    else
        rec<d_normals_t<384>>(c, std::min(nt, share), 0u, (float*)c->vf.p, (const int*)c->nbr.p, (const DevCounters*)c->d_dc, tn1, tord, tslots, holes);
    return F3DS_OK;
}
// stage 2a: seed grid growth
int seg_seed_grid(f3ds_ctx* c) {
    const uint32_t V = c->V;
    float* boxes; const float* vf = (const float*)c->vf.p;
    const uint32_t nchunks = (V + SEED_CHUNK - 1) / SEED_CHUNK;
    ENSURE(c->boxes, float, (size_t)nchunks * 6, boxes);
    rec<d_chunkbox>(c, nchunks, 0u, (const float*)vf, (const DevCounters*)c->d_dc, boxes);
    rec<d_seed_grow>(c, 1u, 0u, (const float*)vf, (const float*)boxes, c->d_dc, c->prm.seed_res, c->d_sgrid);
    return F3DS_OK;
}
// stage 2b: seed cells (sort voxels by cell)
int seg_seed_cells(f3ds_ctx* c, int sort_bits, int max_sdepth) {
    const uint32_t V = c->V;
    uint32_t *ckey, *cell_start, *v0, *v1; uint64_t *k0 = (uint64_t*)c->keys0.p, *k1 = (uint64_t*)c->keys1.p;
    ENSURE(c->vals0, uint32_t, V, v0); ENSURE(c->vals1, uint32_t, V, v1);
    ENSURE(c->ckey, uint32_t, (size_t)V * 3, ckey); ENSURE(c->cell_start, uint32_t, (size_t)V + 1, cell_start);
    rec<d_seed_keys>(c, grid_for(V, 256), 0u, (const float*)c->vf.p, (const DevCounters*)c->d_dc, (const SeedGrid*)c->d_sgrid, ckey, k0, v0);
    int rc = radix_sort(c, k0, v0, k1, v1, V, sort_bits, &c->cks, &c->cvs);
    if (rc) return rc;
    const uint64_t climit = max_sdepth >= 21 ? 0xFFFFFFFFFFFFFFFFull : (1ull << (3 * max_sdepth));
    rec<d_heads>(c, grid_for(V, 256), 0u, (const uint64_t*)c->cks, V, climit, (uint32_t*)c->flags.p, 0);
    if ((rc = scan_u32(c, (const uint32_t*)c->flags.p, (uint32_t*)c->incl.p, V))) return rc;
    rec<d_segstart>(c, grid_for(V, 256), 0u, (const uint64_t*)c->cks, (const uint32_t*)c->flags.p, (const uint32_t*)c->incl.p, V, climit, cell_start, &c->d_dc->n_cells, &c->d_dc->seg_count, 0);
    return F3DS_OK;
}
// stage 2c: nearest voxel per cell, radius filter, kept seeds
int seg_seeds(f3ds_ctx* c) {
    const uint32_t V = c->V, C = c->C;
    uint32_t *sorted_vox, *chvals, *keep; uint64_t* chk; int *seed_orig, *seed_kept;
    const uint32_t ccap = pow2_ge((size_t)C * 2 + 16);
    ENSURE(c->chv, uint32_t, V, sorted_vox); ENSURE(c->chk, uint64_t, ccap, chk); ENSURE(c->chvals, uint32_t, ccap, chvals);
    ENSURE(c->seed_orig, int, C, seed_orig); ENSURE(c->seed_kept, int, C, seed_kept); ENSURE(c->keep, uint32_t, C, keep);
    rec_copy(c, sorted_vox, c->cvs, (size_t)V * 4);      // the sort buffers are reused later
    rec_fill(c, chk, 0xFFFFFFFFu, (size_t)ccap * 8);
    const uint32_t* ckey = (const uint32_t*)c->ckey.p; const uint32_t* cell_start = (const uint32_t*)c->cell_start.p; const float* vf = (const float*)c->vf.p;
    rec<d_cell_hash>(c, grid_for(C, 256), 0u, ckey, (const uint32_t*)sorted_vox, cell_start, (const DevCounters*)c->d_dc, chk, chvals, ccap - 1);
    rec<d_seed_nn>(c, (C + 3u) / 4u, 0u, vf, ckey, (const uint32_t*)sorted_vox, cell_start, (const DevCounters*)c->d_dc, (const SeedGrid*)c->d_sgrid, (const uint64_t*)chk, (const uint32_t*)chvals,
                   ccap - 1, seed_orig);
    rec<d_seed_filter>(c, (C + 3u) / 4u, 0u, vf, ckey, (const uint32_t*)sorted_vox, cell_start, (const DevCounters*)c->d_dc, (const SeedGrid*)c->d_sgrid, (const uint64_t*)chk, (const uint32_t*)chvals, ccap - 1,
                       (const int*)seed_orig, a_radius_sq(c->prm.seed_res), a_min_points(c->prm.seed_res, c->prm.voxel_res), keep);
    int rc = scan_u32(c, keep, (uint32_t*)c->incl.p, C);
    if (rc) return rc;
    rec<d_seed_compact>(c, grid_for(C, 256), 0u, (const int*)seed_orig, (const uint32_t*)keep, (const uint32_t*)c->incl.p, c->d_dc, seed_kept);
    return F3DS_OK;
}
// stage 3: helpers and all label-propagation sweeps
// The state a run of sweeps works on: extract's own (ctx members) or the copy refineSupervoxels continues from
struct SweepBufs {
    Buf *owner, *dist, *hc, *hcount, *hlo, *hhi, *ghost_vox, *ghost_active, *ghost_done, *ghost_head, *ghost_next, *htiles, *htcnt;
    const float* vf;        // voxel features (refine: the copy with the refined normals)
    const int* seeds;       // S0 seed voxels (refine: -1 for helpers erased earlier)
    bool reseed;            // refine: centroids are kept, helpers without a seed stay empty
};
int seg_sweeps_on(f3ds_ctx* c, const SweepBufs& sb) {
    const uint32_t V = c->V, S0 = c->S0;
    const f3ds_params& prm = c->prm;
    uint32_t *owner0, *ownR, *hcount, *hlo, *hhi, *ghost_head, *ghost_next; float *dist0, *hc; unsigned char *R, *ghost_active, *ghost_done; int* ghost_vox;
    ENSURE(*sb.owner, uint32_t, V, owner0); ENSURE(c->ownR, uint32_t, V, ownR); ENSURE(*sb.dist, float, V, dist0);
    ENSURE(c->R, unsigned char, V, R); ENSURE(*sb.hc, float, (size_t)(S0 + 1) * 12, hc); ENSURE(*sb.hcount, uint32_t, S0 + 1, hcount);
    ENSURE(*sb.hlo, uint32_t, S0 + 1, hlo); ENSURE(*sb.hhi, uint32_t, S0 + 1, hhi); ENSURE(*sb.ghost_vox, int, S0 + 1, ghost_vox);
    ENSURE(*sb.ghost_active, unsigned char, S0 + 1, ghost_active); ENSURE(*sb.ghost_done, unsigned char, S0 + 1, ghost_done);
    ENSURE(*sb.ghost_head, uint32_t, V, ghost_head); ENSURE(*sb.ghost_next, uint32_t, S0 + 1, ghost_next);
    const uint32_t T = (V + 63u) / 64u;
    uint32_t *tiles4, *trr, *hD;
    ENSURE(c->tstamp, uint32_t, (size_t)4 * T, tiles4); ENSURE(c->tround, uint32_t, (size_t)(F3DS_R_ROUNDS - 1) * T, trr); ENSURE(c->hdirty, uint32_t, S0 + 1, hD);
    uint32_t *tl, *tcnt; ENSURE(*sb.htiles, uint32_t, (size_t)(S0 + 1) * HT_CAP, tl); ENSURE(*sb.htcnt, uint32_t, S0 + 1, tcnt);
    uint32_t *wl, *wl2, *tmask; ENSURE(c->vwl, uint32_t, V, wl); ENSURE(c->vwl2, uint32_t, V, wl2); ENSURE(c->vtmask, uint32_t, V, tmask);
    rec_fill(c, tiles4, 0u, (size_t)4 * T * 4);
    rec_fill(c, trr, 0u, (size_t)(F3DS_R_ROUNDS - 1) * T * 4);
    rec_fill(c, hD, 0u, (size_t)(S0 + 1) * 4);
    rec_fill(c, owner0, 0u, (size_t)V * 4);
    rec_fill(c, ghost_head, 0u, (size_t)V * 4);
    rec_fill(c, ghost_next, 0u, (size_t)(S0 + 1) * 4);
    { const float fmax = F3DS_FLT_MAX; uint32_t bits; memcpy(&bits, &fmax, 4); rec_fill(c, dist0, bits, (size_t)V * 4); }
    if (sb.reseed) {
        rec<d_reseed_own>(c, grid_for(S0, 256), 0u, sb.seeds, S0, owner0);
        rec<d_reseed_init>(c, grid_for(S0 + 1, 256), 0u, sb.seeds, S0, (const uint32_t*)owner0, ghost_vox, ghost_active, ghost_done, hlo, hhi, hcount, tl, tcnt);
    } else {
        rec<d_helper_own>(c, grid_for(S0, 256), 0u, sb.seeds, S0, owner0);
        rec<d_helper_init>(c, grid_for(S0 + 1, 256), 0u, sb.seeds, S0, (const uint32_t*)owner0, ghost_vox, ghost_active, ghost_done, hlo, hhi, hcount, hc, tl, tcnt);
    }
    const int* nbrT = (const int*)c->nbrT.p; const float* vf = sb.vf;
    SweepFrame a;
    a.sv = SweepView{(int)V, nbrT, vf, owner0, dist0, hc, ghost_head, ghost_next, (const uint32_t*)&c->d_dc->n_ghosts, prm.seed_res, prm.w_normal, prm.w_color, prm.w_spatial, (const int*)c->nbr.p};
    a.R = R; a.ownR = ownR; a.owner_out = owner0; a.dist_out = dist0;
    a.ghost_done = ghost_done; a.ghost_active = ghost_active; a.ghost_vox = ghost_vox; a.ghost_head = ghost_head; a.ghost_next = ghost_next;
    a.hlo = hlo; a.hhi = hhi; a.hcount = hcount; a.hc = hc; a.dc = c->d_dc; a.S0 = S0;
    a.tR0 = tiles4; a.tR1 = tiles4 + T; a.tC0 = tiles4 + 2 * (size_t)T; a.tC1 = tiles4 + 3 * (size_t)T; a.tRr = trr; a.hD = hD; a.T = T; a.tl = tl; a.tcnt = tcnt; a.wl = wl; a.wl2 = wl2; a.tmask = tmask;
    a.thr = g_inc_shift >= 32 ? 0xFFFFFFFFu : (g_inc_shift < 0 ? 0u : V >> g_inc_shift);
    if (!g_sw.sweep_tiles) a.tile_n1 = nullptr;      // development: F3DS_SWEEP_TILES=0 keeps the sweeps on their global-gather path (A/B, tests)
    else a.tile_n1 = (const uint32_t*)c->tile_n1.p;
    a.tile_ord = (const uint32_t*)c->tile_ord.p; a.tile_slots = (const uint32_t*)c->tile_slots.p;
    for (uint32_t t = 0; t < c->res.sweeps; ++t) {
        if (a_sweep_needs_clear(t)) rec_fill(c, R, 0u, V);
        rec<d_sweep_begin>(c, 1u, 0u, a, t);
        // (sweeps 0 and 1 are full by construction -- d_sweep_begin: marking starts after a sweep t >= 1 at the earliest, and a sweep skips tiles only if the one
        // before it was marking --, so their incremental launches would be no-ops on every frame: not recorded)
        const bool can_skip = t >= 2u;
        const bool rounds = g_inc_shift >= 0 && can_skip;
        const uint32_t nrounds = rounds ? g_sw.r_rounds : 0u;      // (F3DS_R_ROUNDS; F3DS_R_ROUNDS_RUN=1|2 lets tests reach the non-convergence fallback of d_sweep_R)
        rec<d_sweep_R_first>(c, grid_for(V, 256), 0u, a, a_sweep_tag(t), t, nrounds);      // pre-pass of a full sweep | round 0 of an incremental one
        for (uint32_t r = 1; r < nrounds; ++r) rec<d_sweep_R_round>(c, grid_for(V, 256), 0u, a, t, r, nrounds);
        for (uint32_t pass = 0; pass < F3DS_R_PASSES; ++pass) rec<d_sweep_R>(c, pass == 0 ? grid_for(V, 256) : 64u, 0u, a, a_sweep_tag(t), t, pass);
        rec<d_sweep_R_tail>(c, 1u, 0u, a, a_sweep_tag(t), t);
        rec<d_sweep_claim>(c, grid_for(V, 256), 0u, a, t);
        rec<d_centroid>(c, grid_for((size_t)S0 * 64u, 256), 0u, a, t);
    }
    return F3DS_OK;
}
int seg_sweeps(f3ds_ctx* c) {
    SweepBufs sb{&c->owner0, &c->dist0, &c->hc, &c->hcount, &c->hlo, &c->hhi, &c->ghost_vox, &c->ghost_active, &c->ghost_done, &c->ghost_head, &c->ghost_next,
                 &c->htiles, &c->htcnt, (const float*)c->vf.p, (const int*)c->seed_kept.p, false};
    return seg_sweeps_on(c, sb);
}
// stage 4a: supervoxel payload rows, adjacency set
int seg_supervoxels(f3ds_ctx* c) {
    const uint32_t V = c->V, S0 = c->S0;
    uint32_t *loff, *rcnt0, *ev0, *ev1; float *rows, *racc0, *rrec0; int* row_voxel; unsigned char* ralive0; uint64_t *ehk, *ek0, *ek1;
    ENSURE(c->loff, uint32_t, S0 + 2, loff);
    int rc = scan_u32(c, (const uint32_t*)c->hcount.p, loff + 1, S0 + 1);     // loff[h+1] = inclusive => loff[h] = exclusive
    if (rc) return rc;
    rec_fill(c, loff, 0u, 4);
    ENSURE(c->rows, float, ((size_t)V + S0 + 1) * 12, rows); ENSURE(c->row_voxel, int, (size_t)V + S0 + 1, row_voxel);      // ghosts add at most S0 rows
    ENSURE(c->racc0, float, (size_t)(S0 + 1) * 12, racc0); ENSURE(c->rcnt0, uint32_t, S0 + 1, rcnt0); ENSURE(c->rrec0, float, (size_t)(S0 + 1) * 16, rrec0);
    ENSURE(c->ralive0, unsigned char, S0 + 1, ralive0);
    rec_fill(c, ralive0, 0u, S0 + 1);
    rec_fill(c, rcnt0, 0u, (size_t)(S0 + 1) * 4);
    rec<d_sv_fill>(c, S0 ? (S0 + 3u) / 4u : 1u, 0u,      // (four helpers per workgroup, one per row of its wave)
                   (const float*)c->vf.p, (const uint32_t*)c->owner0.p, S0, (const uint32_t*)c->hlo.p, (const uint32_t*)c->hhi.p, (const int*)c->ghost_vox.p,
                   (const unsigned char*)c->ghost_active.p, (const uint32_t*)c->hcount.p, (const uint32_t*)loff, (const float*)c->hc.p, rows, row_voxel, racc0, rcnt0, rrec0, ralive0,
                   c->d_dc, (const uint32_t*)c->htiles.p, (const uint32_t*)c->htcnt.p, V);
    const uint32_t ecap = (uint32_t)std::min<uint64_t>((uint64_t)S0 * c->edge_mult + 1024u, 0x7fffffffu);
    const uint32_t ehcap = pow2_ge((size_t)ecap * 2);
    ENSURE(c->ehk, uint64_t, ehcap, ehk);
    ENSURE(c->ekeys0, uint64_t, ecap, ek0); ENSURE(c->ekeys1, uint64_t, ecap, ek1); ENSURE(c->evals0, uint32_t, ecap, ev0); ENSURE(c->evals1, uint32_t, ecap, ev1);
    rec_fill(c, ehk, 0xFFFFFFFFu, (size_t)ehcap * 8);
    rec<d_edges>(c, grid_for(V, 256), 0u, V, S0, (const int*)c->nbrT.p, (const uint32_t*)c->owner0.p, ehk, ehcap - 1, ek0, ecap, c->d_dc);
    rec<d_edges_ghost>(c, grid_for(S0, 256), 0u, V, S0, (const int*)c->ghost_vox.p, (const unsigned char*)c->ghost_active.p, (const int*)c->nbrT.p, (const uint32_t*)c->owner0.p, ehk,
                       ehcap - 1, ek0, ecap, c->d_dc);
    return F3DS_OK;
}
// stage 4b: sorted edge list
int seg_edge_sort(f3ds_ctx* c, int sort_bits) {
    const uint32_t E = c->E, S0 = c->S0;
    uint32_t *ea0, *eb0; ENSURE(c->ea0, uint32_t, E, ea0); ENSURE(c->eb0, uint32_t, E, eb0);
    uint32_t* evs;
    { uint32_t* h; ENSURE(c->hist, uint32_t, (size_t)RS_BINS * (((size_t)E * 2 + RS_TILE - 1) / RS_TILE + 1), h); }      // also holds the 2E-delta sort of seg_cluster_front, recorded before this one is flushed
    rec<d_iota>(c, grid_for(E, 256), 0u, (uint32_t*)c->evals0.p, E);
    int rc = radix_sort(c, (uint64_t*)c->ekeys0.p, (uint32_t*)c->evals0.p, (uint64_t*)c->ekeys1.p, (uint32_t*)c->evals1.p, E, sort_bits, &c->eks, &evs);
    if (rc) return rc;
    rec<d_edge_init>(c, grid_for(E, 256), 0u, (const uint64_t*)c->eks, E, S0, ea0, eb0);
    return F3DS_OK;
}
// d_merge_il_t<NW, RES>: LDS a frame with E adjacencies needs (merge_il_offsets, f3ds_kernels.inc) and whether the kernel can take it
bool merge_il_layout(uint32_t E, uint32_t S0, int nw, int res, MergeLds* xl) {      // res: 2 = endpoints + keys in LDS, 0 = both in global memory
    memset(xl, 0, sizeof *xl);
    if ((uint64_t)E * (res == 2 ? 10u : 2u) > (1u << 22)) return false;              // (far beyond what fits: keep the 32-bit offsets below honest)
    const MergeIlLayout L = merge_il_offsets(E, nw, res);
    xl->Ecap = L.Ecap; xl->NGcap = L.NGcap; xl->lds_bytes = L.total; xl->keys_in_lds = res;
    xl->spec = g_sw.merge_spec ? 1 : 0;      // (the speculative second merge of an epoch; F3DS_MERGE_SPEC=0 switches it off)
    return L.total <= MC_LDS_LIMIT && S0 <= 65534u;
}
// merge kernel of a batch: MK_GLOBAL (d_merge, everything in HBM: any size) or MK_IL + (8 waves ? 0 : 2) + (res == 2 ? 0 : 1)
enum MergeKind { MK_GLOBAL = 0, MK_IL = 1 };
inline int mk_waves(int kind) { return (kind - MK_IL) >= 2 ? 4 : 8; }
inline int mk_res(int kind) { return (kind - MK_IL) % 2 ? 0 : 2; }
// Which merge kernel a batch runs (one dispatch for all its frames): 8 waves per frame with keys and endpoints in LDS -- the shortest loop, what a lone frame or
// a lone call wants.  A call of 16 frames or more that shares the device with other batch calls takes the 4-wave layout instead: its loop is longer, but a
// workgroup holds one wave slot and 250 registers per SIMD instead of two and 500, and the other calls' wide kernels run on the units the merge loops sit on
// (DESIGN.md 4i).  Results are identical whatever runs.  F3DS_MERGE_KEYS=lds|global says where the per-edge arrays live, F3DS_MERGE_NW=4|8 the width.
// A frame whose arrays fit neither way, or with more than 65534 seeds, takes d_merge.  F3DS_FORCE_GLOBAL_MERGE forces it (tests).
int choose_merge_kind(const std::vector<f3ds_ctx*>& fr, bool force_global) {
    if (force_global || g_sw.force_global_merge) return MK_GLOBAL;
    const bool e_keys = g_sw.merge_keys >= 0;
    const bool shared = fr.size() >= 16 && g_batch_calls[fr[0]->device & 15].load(std::memory_order_relaxed) >= 2;
    const int nw = g_sw.merge_nw ? (g_sw.merge_nw == 8 ? 8 : 4) : (shared ? 4 : 8);
    const int first = e_keys ? (g_sw.merge_keys == 2 ? 2 : 0) : (shared ? g_merge_shared_res : 2);
    for (int res = first; res >= (e_keys ? first : 0); res -= 2) {
        bool ok = true;
        for (f3ds_ctx* c : fr) { MergeLds t; if (!merge_il_layout(c->E, c->S0, nw, res, &t)) { ok = false; break; } }
        if (ok) return MK_IL + (nw == 8 ? 0 : 2) + (res == 2 ? 0 : 1);
    }
    return MK_GLOBAL;
}
// stage 4c: Clustering::cluster(threshold) up to the merge loop: working copies, deltas, lambda / cdf, weights
int seg_cluster_front(f3ds_ctx* c, const f3ds_params* prm, int kind) {
    const bool use_lds = kind != MK_GLOBAL;
    const uint32_t S0 = c->S0, E = c->E;
    // main(): set_merging / set_lambda / set_bins_num (src/supervoxel_clustering.cpp:415-423)
    float lambda = 0.5f; int bins = 500;
    if (prm->merging == F3DS_MANUAL_LAMBDA && prm->lambda != 0) { if (prm->lambda < 0 || prm->lambda > 1) return F3DS_ERR_RANGE; lambda = prm->lambda; }
    if (prm->merging == F3DS_EQUALIZATION && prm->bins != 0) { if (prm->bins < 0) return F3DS_ERR_RANGE; bins = (short)prm->bins; }
    if (prm->merging < 0 || prm->merging > 2 || prm->color_metric < 0 || prm->color_metric > 1 || prm->geom_metric < 0 || prm->geom_metric > 1) return F3DS_ERR_ARG;
    MergeDev m;
    memset(&m, 0, sizeof m);
    m.E = E; m.S0 = S0; m.threshold = prm->threshold; m.dc = c->d_dc;
    { const uint64_t cap = (uint64_t)E * c->ev_mult + 4096u; m.ev_cap = cap > 0x7fffffffull ? 0x7fffffffu : (uint32_t)cap; }      // weight-history events: grown on demand (run_cluster)
    ENSURE(c->ea, uint32_t, E, m.ea); ENSURE(c->eb, uint32_t, E, m.eb); ENSURE(c->ew, float, ((size_t)E + 2047) & ~(size_t)2047, m.ew); /* d_merge: weights; d_merge_cw_t<., 0>: endpoints */ ENSURE(c->eku, uint32_t, ((size_t)E + 2047) & ~(size_t)2047, m.eku);      // (Ecap of the widest merge kernel)
    ENSURE(c->ehist, int, E, m.ehist); ENSURE(c->ealive, unsigned char, E, m.ealive);
    ENSURE(c->ev_epoch, uint32_t, m.ev_cap, m.ev_epoch); ENSURE(c->ev_key, uint32_t, m.ev_cap, m.ev_key); ENSURE(c->ev_prev, int, m.ev_cap, m.ev_prev);
    ENSURE(c->racc, float, (size_t)(S0 + 1) * 12, m.racc); ENSURE(c->rrec, float, (size_t)(S0 + 1) * 16, m.rrec);
    ENSURE(c->rcnt, uint32_t, S0 + 1, m.rcnt); ENSURE(c->ralive, unsigned char, S0 + 1, m.ralive);
    ENSURE(c->rhead, uint32_t, S0 + 1, m.rhead); ENSURE(c->rtail, uint32_t, S0 + 1, m.rtail); ENSURE(c->lnext, uint32_t, S0 + 1, m.lnext);
    ENSURE(c->parent, uint32_t, S0 + 1, m.parent); ENSURE(c->markA, uint32_t, S0 + 1, m.markA); ENSURE(c->markB, uint32_t, S0 + 1, m.markB);
    ENSURE(c->tl, uint32_t, E, m.tl); ENSURE(c->merges, uint32_t, (size_t)(S0 + 1) * 3, m.merges);
    m.loff = (const uint32_t*)c->loff.p; m.llen = (const uint32_t*)c->hcount.p; m.rows = (const float*)c->rows.p;
    float* deltas; ENSURE(c->deltas, float, (size_t)E * 2, deltas);
    // working copies of the supervoxel state (a second cluster() call starts from the same initial state)
    rec_copy(c, m.racc, c->racc0.p, (size_t)(S0 + 1) * 12 * 4);
    rec_copy(c, m.rrec, c->rrec0.p, (size_t)(S0 + 1) * 16 * 4);
    rec_copy(c, m.rcnt, c->rcnt0.p, (size_t)(S0 + 1) * 4);
    rec_copy(c, m.ralive, c->ralive0.p, (size_t)(S0 + 1));
    rec_copy(c, m.ea, c->ea0.p, (size_t)E * 4);
    rec_copy(c, m.eb, c->eb0.p, (size_t)E * 4);
    MergeLds xl;
    merge_il_layout(E, S0, use_lds ? mk_waves(kind) : 8, use_lds ? mk_res(kind) : 0, &xl);
    uint32_t logS = 1; while ((1u << logS) < S0 + 2u) ++logS;
    xl.pool_cap = (S0 + 1u) * (4u * logS + 8u) * c->pool_mult;
    ENSURE(c->pool, uint2, xl.pool_cap, xl.pool); ENSURE(c->rstart, uint32_t, S0 + 1, xl.rstart); ENSURE(c->rnleaf, uint32_t, S0 + 1, xl.rnleaf);
    ENSURE(c->rcap, uint32_t, S0 + 1, xl.rcap);
    // incident-edge lists of the regions (d_inc_build): the initial lists take 2 E entries, a merge whose touched list outgrows a's segment takes a fresh one
    // (only the incident-list kernels have them: d_merge, the all-global fallback of the large-E scenes, scans the edge arrays.  What the loop can need is bounded: a
    // merge that leaves its segment takes at most nt + nt / 2 + 4 fresh entries, nt <= MC_TL_CAP, and there are at most S0 - 1 merges)
    if (use_lds) {
        const uint64_t bound = 2ull * E + (uint64_t)(S0 ? S0 - 1u : 0u) * (MC_TL_CAP + MC_TL_CAP / 2u + 4u) + 1024u;
        uint64_t cap = 2ull * E + (uint64_t)g_sw.ilist_slack * E * c->ilist_mult + (g_sw.ilist_slack < 32u ? 16u : 1024u);
        if (cap > bound) cap = bound;
        xl.ilist_cap = cap > 0x7fffffffull ? 0x7fffffffu : (uint32_t)cap;
        ENSURE(c->ilist, uint32_t, xl.ilist_cap, xl.ilist); ENSURE(c->istart, uint32_t, S0 + 1, xl.istart); ENSURE(c->ilen, uint32_t, S0 + 1, xl.ilen); ENSURE(c->icap, uint32_t, S0 + 1, xl.icap);
    }
    if (use_lds) rec<d_inc_build>(c, 1u, 0u, E, S0, (const uint32_t*)m.ea, (const uint32_t*)m.eb, xl.istart, xl.ilen, xl.icap, xl.ilist);
    xl.stop_key = (prm->threshold != prm->threshold) ? 0u : n_weight_key(prm->threshold);
    c->merge_in_lds = use_lds; c->merge_kind = kind;
    rec<d_region_reset>(c, grid_for(S0 + 1, 256), 0u, S0, (const uint32_t*)c->hcount.p, m.rhead, m.rtail, m.lnext, m.parent, m.markA, m.markB, xl.pool, xl.rstart, xl.rnleaf, xl.rcap, (const uint32_t*)c->loff.p);
    m.mp.color_metric = prm->color_metric; m.mp.geom_metric = prm->geom_metric; m.mp.merging = prm->merging; m.mp.lambda = lambda; m.mp.bins = bins;
    uint64_t *sk0 = nullptr, *sk1 = nullptr; uint32_t *sv0 = nullptr, *sv1 = nullptr;
    if (prm->merging == F3DS_ADAPTIVE_LAMBDA) {
        ENSURE(c->skeys0, uint64_t, (size_t)E * 2, sk0); ENSURE(c->skeys1, uint64_t, (size_t)E * 2, sk1);
        ENSURE(c->svals0, uint32_t, (size_t)E * 2, sv0); ENSURE(c->svals1, uint32_t, (size_t)E * 2, sv1);
    }
    rec<d_edge_deltas>(c, grid_for(E, 256), 0u, E, (const uint32_t*)m.ea, (const uint32_t*)m.eb, (const float*)m.rrec, prm->color_metric, prm->geom_metric, deltas, sk0, sv0);
    if (prm->merging == F3DS_ADAPTIVE_LAMBDA) {
        uint64_t* ks; uint32_t* vs;
        int rc = radix_sort(c, sk0, sv0, sk1, sv1, E * 2u, 33, &ks, &vs);
        if (rc) return rc;
        rec<d_lambda>(c, 1u, 0u, E, (const float*)deltas, (const uint32_t*)vs, c->d_dc);
    } else if (prm->merging == F3DS_EQUALIZATION) {
        uint32_t* hist; float* cdf;
        ENSURE(c->cdf_hist, uint32_t, (size_t)2 * (bins > 0 ? bins : 1), hist); ENSURE(c->cdf, float, (size_t)2 * (bins > 0 ? bins : 1), cdf);
        rec_fill(c, hist, 0u, (size_t)2 * (bins > 0 ? bins : 1) * 4);
        rec<d_cdf_hist>(c, grid_for((size_t)E * 2, 256), 0u, E, (const float*)deltas, bins, hist, c->d_dc);
        rec<d_cdf_scan>(c, 1u, 0u, E, bins, (const uint32_t*)hist, cdf);
        m.mp.cdf_c = cdf; m.mp.cdf_g = cdf + bins;
    }
    rec<d_edge_weights>(c, grid_for(E, 256), 0u, m, (const float*)deltas);
    c->mdev = m; c->mlds = xl; c->host_lambda = lambda;
    return F3DS_OK;
}
// stage 5: the merge loop, one workgroup per frame
int seg_merge(f3ds_ctx* c) {
#ifdef F3DS_WHATIF      // make WHATIF=1 only: the stand-in changes the labels, so the default library does not contain it
    if (const char* e = dev_getenv("F3DS_FAKE_MERGE")) {      // experiment, timing only: see d_fake_merge
        unsigned us = 30000, waves = 1; sscanf(e, "%u,%u", &us, &waves);
        const char* l = dev_getenv("F3DS_FAKE_MERGE_LDS");
        rec<d_fake_merge>(c, 1u, l ? (uint32_t)atoi(l) * 1024u : c->mlds.lds_bytes, c->mdev, (uint32_t)us, (uint32_t)waves);
        return F3DS_OK;
    }
#endif
    switch (c->merge_kind) {
        case MK_IL + 0: rec<d_merge_il_t<8, 2>>(c, 1u, c->mlds.lds_bytes, c->mdev, c->mlds); break;
        case MK_IL + 1: rec<d_merge_il_t<8, 0>>(c, 1u, c->mlds.lds_bytes, c->mdev, c->mlds); break;
        case MK_IL + 2: rec<d_merge_il_t<4, 2>>(c, 1u, c->mlds.lds_bytes, c->mdev, c->mlds); break;
        case MK_IL + 3: rec<d_merge_il_t<4, 0>>(c, 1u, c->mlds.lds_bytes, c->mdev, c->mlds); break;
        default: rec<d_merge>(c, 1u, 0u, c->mdev);
    }
    return F3DS_OK;
}
// stage 6: region ids (ascending surviving label) and per-point labels: one kernel (rank table in LDS), or two when some frame of
// the batch has more supervoxels than the table holds (c->relabel_lds, decided for the whole batch in run_cluster)
int seg_labels(f3ds_ctx* c) {
    const uint32_t S0 = c->S0, n = c->n;
    const MergeDev& m = c->mdev;
    uint32_t *root, *rincl, *d_labels;
    ENSURE(c->root, uint32_t, S0 + 1, root); ENSURE(c->rincl, uint32_t, S0 + 1, rincl);
    if (c->user_labels) d_labels = c->user_labels;      // a device output buffer is written in place
    else ENSURE(c->labels, uint32_t, n, d_labels);
    if (c->relabel_lds) {
        // (every workgroup builds the table: a lone frame does not get more workgroups than it has 4096-point slices)
        const uint32_t gx = std::min(grid_for(n, 256), grid_wide(n, 4096));
        rec<d_relabel>(c, gx, (S0 + 1u) * 4u, n, (const int*)c->pt_voxel.p, (const uint32_t*)c->owner0.p, S0, (const uint32_t*)m.parent, (const unsigned char*)m.ralive, root, rincl, d_labels, c->d_dc);
    } else {
        uint32_t* rank; ENSURE(c->rrank, uint32_t, S0 + 1, rank);
        rec<d_region_ids>(c, 1u, 0u, S0, (const uint32_t*)m.parent, (const unsigned char*)m.ralive, rank, root, rincl, c->d_dc);
        rec<d_point_labels>(c, grid_for(n, 256), 0u, n, (const int*)c->pt_voxel.p, (const uint32_t*)c->owner0.p, (const uint32_t*)rank, d_labels);
    }
    return F3DS_OK;
}

// labels of a frame that has no voxels at all
int finish_empty(f3ds_ctx* c, hipStream_t st, uint32_t* point_labels, int labels_on_device) {
    uint32_t* d_labels;
    ENSURE(c->labels, uint32_t, c->n ? c->n : 1, d_labels);
    if (c->n) {
        HIPCHECK(hipMemsetAsync(d_labels, 0xFF, (size_t)c->n * 4, st));
        if (point_labels) HIPCHECK(hipMemcpyAsync(point_labels, d_labels, (size_t)c->n * 4, labels_on_device ? hipMemcpyDefault : hipMemcpyDeviceToHost, st));
    }
    c->have_frame = false; c->live = false;
    return F3DS_OK;
}

// Streams for batch calls.  HIP multiplexes streams onto a few hardware queues (4 by default) in creation order, and two
// batches whose streams share a hardware queue run one after the other (measured: with 576 contexts created in a row, two
// of bench.py's three concurrent batches landed in one queue).  A batch therefore does not run on its first context's
// stream but on one of four streams per device (GPU_MAX_HW_QUEUES of them if that is set) created back to back --
// different queues -- and held for the call.
// ONE stream per device for the host <-> device copies of ALL calls (F3DS_COPY_STREAM=0: every call copies on its own stream, as until round 4).  On this platform a
// host-to-device and a device-to-host copy that run at the same time get 16 GB/s each where either alone gets 55 (tools/pcie_bw.py); with six calls in flight the uploads of
// one call (16 MB per frame) kept meeting the label downloads of another (4 MB per frame).  Through one stream the link carries one copy at a time at full rate.  A call's
// downloads are queued only once its labels exist (the host thread waits for the event first), so the stream never sits blocked behind unfinished compute.
struct CopyStream { std::mutex m; hipStream_t s = nullptr; };
CopyStream g_copy_stream[16][2];      // [1]: a second stream for the downloads (development: F3DS_COPY_STREAM=2, both directions of the link at once)
hipStream_t copy_stream_of(int device, int which = 0) {
    CopyStream& c = g_copy_stream[device & 15][which & 1];
    std::lock_guard<std::mutex> lk(c.m);
    if (!c.s && hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking) != hipSuccess) c.s = nullptr;
    return c.s;
}
struct BatchStreamPool { std::mutex m; hipStream_t s[8] = {}; bool busy[8] = {}; };
const int g_batch_stream_count = [] { const char* e = getenv("GPU_MAX_HW_QUEUES"); int n = e ? atoi(e) : 4; return n < 1 ? 1 : (n > 8 ? 8 : n); }();      // one per hardware queue HIP will use
BatchStreamPool g_batch_streams[16];
struct BatchStreamLease {
    int dev = -1, slot = -1;
    hipStream_t acquire(int device) {
        BatchStreamPool& p = g_batch_streams[device & 15];
        std::lock_guard<std::mutex> lk(p.m);
        if (!p.s[0]) for (int i = 0; i < g_batch_stream_count; ++i) if (hipStreamCreateWithFlags(&p.s[i], hipStreamNonBlocking) != hipSuccess) { p.s[i] = nullptr; return nullptr; }
        for (int i = 0; i < g_batch_stream_count; ++i) if (!p.busy[i] && p.s[i]) { p.busy[i] = true; dev = device & 15; slot = i; return p.s[i]; }
        return nullptr;      // more batches at once on this device than queues: the caller's own stream
    }
    ~BatchStreamLease() { if (slot >= 0) { std::lock_guard<std::mutex> lk(g_batch_streams[dev].m); g_batch_streams[dev].busy[slot] = false; } }
};

// run `fn` (a per-frame recorder) on every live frame; a failing frame fails the batch
template <class F>
int for_frames(Batch& b, F&& fn) {
    for (f3ds_ctx* c : b.fr) { int rc = fn(c); if (rc) return rc; }
    return F3DS_OK;
}
void stage_mark(Batch& b, int i) { (void)hipEventRecord(b.owner->ev[i], b.st); }

// device address of a pinned host buffer (nullptr for pageable memory, which the device cannot reach)
static uint32_t* pinned_device_alias(uint32_t* host) {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }      // (pageable memory: "invalid value", and the error must not stick)
    if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
    return (uint32_t*)at.devicePointer;
}

// cluster stage for the live frames (also the whole of f3ds_recluster)
int run_cluster(Batch& b, const f3ds_params* prm, uint32_t* const* labels_of, const std::vector<int>& index_of, int labels_on_device, bool force_global = false) {
    const int kind = choose_merge_kind(b.fr, force_global);      // one kernel for the whole batch
    const bool all_lds = kind != MK_GLOBAL;
    int rc = for_frames(b, [&](f3ds_ctx* c) { return seg_cluster_front(c, prm, kind); });
    if (rc || (rc = flush(b))) return rc;
    stage_mark(b, 5);
    if ((rc = for_frames(b, seg_merge)) || (rc = flush(b))) return rc;
    stage_mark(b, 6);
    for (size_t i = 0; i < b.fr.size(); ++i) {
        uint32_t* out = labels_of ? labels_of[index_of[i]] : nullptr;
        // F3DS_DIRECT_LABELS=1 (experiment, measured and not the default): a host label buffer that is pinned (hipHostMalloc / hipHostRegister) is written by the relabel kernel
        // itself through its device address instead of a staging buffer + device-to-host copy.  With six calls in flight the pinned-host-in / host-out rate of bench.py
        // drops from 2 060 to 1 830 Mpoints/s: the kernel holds its workgroups while its stores trickle up the link.
        if (out && !labels_on_device && g_sw.direct_labels) out = pinned_device_alias(out);
        else if (!labels_on_device) out = nullptr;
        b.fr[i]->user_labels = out;
    }
    {
        const uint32_t cap = g_sw.relabel_lds_cap >= 0 ? (uint32_t)g_sw.relabel_lds_cap : RL_LDS_CAP;      // (tests: 0 forces the two-kernel form)
        bool fits = true;
        for (f3ds_ctx* c : b.fr) if (c->S0 + 1u > cap) fits = false;
        for (f3ds_ctx* c : b.fr) c->relabel_lds = fits;
    }
    if ((rc = for_frames(b, seg_labels)) || (rc = flush(b))) return rc;
    stage_mark(b, 7);
    {
        bool want = false;
        for (size_t i = 0; i < b.fr.size(); ++i) if (labels_of && labels_of[index_of[i]] && !b.fr[i]->user_labels && b.fr[i]->n) want = true;
        hipStream_t dl = (want && !labels_on_device && g_sw.copy_stream) ? copy_stream_of(b.fr[0]->device, g_sw.copy_duplex ? 1 : 0) : nullptr;
        if (dl) {      // queued on the copy stream only once the labels exist: that stream must never sit blocked behind unfinished compute
            for (int k = 1; k < 3; ++k) if (!b.owner->ev_copy[k]) HIPCHECK(hipEventCreateWithFlags(&b.owner->ev_copy[k], hipEventDisableTiming));
            HIPCHECK(hipEventRecord(b.owner->ev_copy[1], b.st));
            { const double t0 = now_ms(); HIPCHECK(hipEventSynchronize(b.owner->ev_copy[1])); g_t_wait += now_ms() - t0; }
        }
        for (size_t i = 0; i < b.fr.size(); ++i) {
            f3ds_ctx* c = b.fr[i];
            uint32_t* out = labels_of ? labels_of[index_of[i]] : nullptr;
            if (out && !c->user_labels && c->n) HIPCHECK(hipMemcpyAsync(out, c->labels.p, (size_t)c->n * 4, hipMemcpyDeviceToHost, dl ? dl : b.st));
            c->user_labels = nullptr;
        }
        if (dl) HIPCHECK(hipEventRecord(b.owner->ev_copy[2], dl));
        if ((rc = flush_sync(b))) return rc;
        if (dl) { const double t0 = now_ms(); HIPCHECK(hipEventSynchronize(b.owner->ev_copy[2])); g_t_wait += now_ms() - t0; }
    }
    {
        // a frame whose merge loop re-weights more edges than its event arrays hold (huge regions of tiny supervoxels)
        // runs the stage again -- it starts from the untouched supervoxel state -- with four times the room
        bool again = false;
        for (f3ds_ctx* c : b.fr) {
            if (c->h_dc->ev_overflow == 1 && c->ev_mult < 16384u) { c->ev_mult *= 4u; again = true; }
            if (c->h_dc->ev_overflow == 2 && c->pool_mult < 64u) { c->pool_mult *= 4u; again = true; }
            if (c->h_dc->ev_overflow == 4 && c->ilist_mult < 4096u) { c->ilist_mult *= 4u; again = true; }
        }
        if (again) {
            if (g_sw.trace_err) fprintf(stderr, "f3ds: merge stage of %zu frames runs again with more history / leaf-pool room\n", b.fr.size());
            for (f3ds_ctx* c : b.fr) {
                c->h_dc->error = 0; c->h_dc->ev_overflow = 0;
                HIPCHECK(hipMemsetAsync(&c->d_dc->error, 0, sizeof(int), b.st)); HIPCHECK(hipMemsetAsync(&c->d_dc->ev_overflow, 0, sizeof(int), b.st));
            }
            return run_cluster(b, prm, labels_of, index_of, labels_on_device, force_global);
        }
    }
    if (all_lds) {
        // a merge whose two regions touch more than ML_TL_CAP edges does not fit the LDS kernel's lists: the stage
        // (it starts from the untouched supervoxel state) runs again with the global-memory kernel
        bool again = false;
        for (f3ds_ctx* c : b.fr) if (c->h_dc->error == F3DS_ERR_UNSUPPORTED) again = true;
        if (again) {
            if (g_sw.trace_err) fprintf(stderr, "f3ds: merge stage of %zu frames runs again with the global-memory kernel\n", b.fr.size());
            for (f3ds_ctx* c : b.fr) { c->h_dc->error = 0; HIPCHECK(hipMemsetAsync(&c->d_dc->error, 0, sizeof(int), b.st)); }
            return run_cluster(b, prm, labels_of, index_of, labels_on_device, true);
        }
    }
    for (f3ds_ctx* c : b.fr) {
        if (c->h_dc->error) return trace_err(c->h_dc->error, "merge", c);
        c->res.n_merges = c->h_dc->n_merges; c->res.n_regions = c->h_dc->n_regions;
        c->res.lambda = prm->merging == F3DS_ADAPTIVE_LAMBDA ? (c->E ? c->h_dc->lambda : __builtin_nanf("")) : c->host_lambda;
        c->prm.color_metric = prm->color_metric; c->prm.geom_metric = prm->geom_metric; c->prm.merging = prm->merging;
        c->prm.lambda = prm->lambda; c->prm.bins = prm->bins; c->prm.threshold = prm->threshold;
        c->have_frame = true;
    }
    return F3DS_OK;
}

}  // namespace

extern "C" {

const char* f3ds_last_hip_error(void) { return g_last_hip_error.c_str(); }

int f3ds_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int f3ds_create(int device, f3ds_ctx** out) {
    if (!out) return F3DS_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return F3DS_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return F3DS_ERR_ARG;
    HIPCHECK(hipSetDevice(device));
    f3ds_ctx* c = new f3ds_ctx;
    c->device = device;
    if (const char* e = dev_getenv("F3DS_EDGE_MULT")) { const int v = atoi(e); if (v >= 1 && v <= 32) c->edge_mult = (uint32_t)v; }      // tests: start with a short adjacency list
    const int rc = [c]() -> int {
        HIPCHECK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        for (auto& e : c->ev) HIPCHECK(hipEventCreate(&e));
        HIPCHECK(hipMalloc((void**)&c->d_dc, sizeof(DevCounters)));
        HIPCHECK(hipHostMalloc((void**)&c->h_dc, sizeof(DevCounters), hipHostMallocDefault));
        HIPCHECK(hipMalloc((void**)&c->d_grid, sizeof(GridInfo)));
        HIPCHECK(hipHostMalloc((void**)&c->h_grid, sizeof(GridInfo), hipHostMallocDefault));
        HIPCHECK(hipMalloc((void**)&c->d_sgrid, sizeof(SeedGrid)));
        return F3DS_OK;
    }();
    if (rc) { c->stream = c->own_stream; f3ds_destroy(c); return rc; }      // a half-built context is released, not leaked
    memset(c->h_grid, 0, sizeof(GridInfo));
    memset(&c->res, 0, sizeof c->res);
    *out = c;
    return F3DS_OK;
}

void f3ds_destroy(f3ds_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    Buf* bufs = &c->pts;
    const size_t nb = (reinterpret_cast<char*>(&c->rincl) - reinterpret_cast<char*>(&c->pts)) / sizeof(Buf) + 1;
    for (size_t i = 0; i < nb; ++i) if (bufs[i].p) (void)hipFree(bufs[i].p);
    if (c->d_dc) (void)hipFree(c->d_dc);
    if (c->h_dc) (void)hipHostFree(c->h_dc);
    if (c->d_grid) (void)hipFree(c->d_grid);
    if (c->h_grid) (void)hipHostFree(c->h_grid);
    if (c->d_sgrid) (void)hipFree(c->d_sgrid);
    for (int k = 0; k < 2; ++k) { if (c->h_args[k]) (void)hipHostFree(c->h_args[k]); if (c->d_args[k]) (void)hipFree(c->d_args[k]); if (c->ev_args[k]) (void)hipEventDestroy(c->ev_args[k]); }
    for (auto& e : c->ev_copy) if (e) (void)hipEventDestroy(e);
    if (c->d_dcblk) (void)hipFree(c->d_dcblk);
    if (c->h_dcblk) (void)hipHostFree(c->h_dcblk);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int f3ds_set_stream(f3ds_ctx* c, void* hip_stream) {
    if (!c) return F3DS_ERR_ARG;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return F3DS_OK;
}

int f3ds_segment(f3ds_ctx* c, const void* points, size_t n_, int points_on_device, const f3ds_params* prm, uint32_t* point_labels,
                 int labels_on_device, f3ds_result* result) {
    const void* pp[1] = {points}; const size_t cnt[1] = {n_}; uint32_t* lp[1] = {point_labels};
    return f3ds_segment_batch(&c, 1, pp, cnt, points_on_device, prm, lp, labels_on_device, result);
}

// A batch of independent frames (BASELINE.json config 5: 8 frames per GPU) walks the stages in
// lockstep on the first context's stream: per stage every frame records its kernel calls, flush()
// turns them into batched dispatches (grid.y = frame), and the few host decisions (octree depth,
// voxel / seed / edge counts) are taken for all frames at one synchronisation point per stage.
int f3ds_segment_batch(f3ds_ctx** ctxs, int nctx, const void* const* points, const size_t* counts, int points_on_device, const f3ds_params* prm,
                       uint32_t* const* point_labels, int labels_on_device, f3ds_result* results) {
    if (!ctxs || nctx <= 0 || !points || !counts || !prm) return F3DS_ERR_ARG;
    if (!(prm->voxel_res > 0) || !(prm->seed_res > 0)) return F3DS_ERR_ARG;
    for (int i = 0; i < nctx; ++i) if (!ctxs[i] || (!points[i] && counts[i]) || counts[i] > 0x7fffffffull || ctxs[i]->device != ctxs[0]->device) return F3DS_ERR_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    g_t_wait = 0; g_t_launch = 0;
    g_sw.read();
    HIPCHECK(hipSetDevice(ctxs[0]->device));
    BatchCallCount in_flight(ctxs[0]->device);
    Batch b;
    b.owner = ctxs[0]; b.st = ctxs[0]->stream;
    BatchStreamLease lease;
    if (nctx > 1 && ctxs[0]->stream == ctxs[0]->own_stream && !g_sw.no_stream_pool) { hipStream_t ps = lease.acquire(ctxs[0]->device); if (ps) b.st = ps; }
    g_grid_cap = grid_cap_for_batch(nctx); g_batch_frames = nctx;
    std::vector<int> index_of;
    const int max_depth = (int)(1.8f * prm->seed_res / prm->voxel_res);      // [PCL-recall] SupervoxelClustering::extract
    const uint32_t sweeps = max_depth > 1 ? (uint32_t)(max_depth - 1) : 0u;
    stage_mark(b, 0);
    hipStream_t up_stream = (!points_on_device && g_sw.copy_stream) ? copy_stream_of(ctxs[0]->device) : nullptr;
    bool uploaded = false;
    for (int i = 0; i < nctx; ++i) {
        f3ds_ctx* c = ctxs[i];
        c->cmds.clear(); c->blob.clear(); c->pend.clear(); c->ops.n = 0; c->ops_grid = 0;
        c->have_frame = false; c->live = true; c->rc = 0; c->refined_itr = -1;
        c->user_mode = false; c->user_label.clear(); c->user_row.clear();
        { const int prc = pregrow_scratch(c); if (prc) return prc; }
        c->prm = *prm; c->n = (uint32_t)counts[i]; c->V = c->C = c->S0 = c->E = 0;
        memset(&c->res, 0, sizeof c->res);
        c->res.n_points = c->n; c->res.sweeps = sweeps;
        c->fa = FrameArgs{prm->use_transform, prm->fold_negative_z, prm->leaf_order, prm->voxel_res, prm->seed_res, prm->w_color, prm->w_spatial, prm->w_normal};
        if (points_on_device) c->d_pts = (const P16*)points[i];
        else {
            P16* up; ENSURE(c->pts, P16, c->n ? c->n : 1, up);
            if (c->n) { HIPCHECK(hipMemcpyAsync(up, points[i], (size_t)c->n * 16, hipMemcpyHostToDevice, up_stream ? up_stream : b.st)); uploaded = true; }
            c->d_pts = up;
        }
        b.fr.push_back(c); index_of.push_back(i);
    }
    if (up_stream && uploaded) {      // the call's kernels wait for its uploads, which went through the device's copy stream
        if (!b.owner->ev_copy[0]) HIPCHECK(hipEventCreateWithFlags(&b.owner->ev_copy[0], hipEventDisableTiming));
        HIPCHECK(hipEventRecord(b.owner->ev_copy[0], up_stream));
        HIPCHECK(hipStreamWaitEvent(b.st, b.owner->ev_copy[0], 0));
    }
    auto drop_dead = [&](auto&& dead) -> int {      // frames without voxels leave the batch with all labels = F3DS_NO_LABEL
        std::vector<f3ds_ctx*> keep; std::vector<int> keep_idx;
        for (size_t i = 0; i < b.fr.size(); ++i) {
            f3ds_ctx* c = b.fr[i];
            if (dead(c)) { int rc = finish_empty(c, b.st, point_labels ? point_labels[index_of[i]] : nullptr, labels_on_device); if (rc) return rc; }
            else { keep.push_back(c); keep_idx.push_back(index_of[i]); }
        }
        b.fr.swap(keep); index_of.swap(keep_idx);
        return F3DS_OK;
    };
    int rc;
    // ---- stage 0: voxelise
    if ((rc = for_frames(b, seg_bbox)) || (rc = flush_sync(b))) return rc;
    for (f3ds_ctx* c : b.fr) {
        c->res.n_finite = c->h_dc->n_finite;
        if (c->h_dc->error) return c->h_dc->error;
        c->res.octree_depth = (uint32_t)c->h_dc->depth;
    }
    if ((rc = drop_dead([](f3ds_ctx* c) { return c->h_dc->grid_empty || c->n == 0; }))) return rc;
    int maxd = 0;
    for (f3ds_ctx* c : b.fr) if (c->h_dc->depth > maxd) maxd = c->h_dc->depth;
    // Stage 0b/0c.  Default: the hash path (only the voxels are sorted).  The sort path (every point's key through a three-pass radix sort) remains for frames with
    // very dense voxels, for lone frames (below) and behind F3DS_VOX_TILES=0.
    // (a lone frame takes the sort path: two of the tile path's kernels are one workgroup per frame -- 2.4 ms on a lone frame's critical path, nothing in a batch)
    bool hashed = g_sw.vox_hash && (nctx >= 4 || g_sw.vox_tiles_forced);
    for (f3ds_ctx* c : b.fr) {
        c->vox_hashed = false;
        if (c->vox_dense && hashed) { if (c->vox_dense_wait) c->vox_dense_wait--; else c->vox_dense = false; }      // (counted in calls that would have taken the tile path)
    }
    for (f3ds_ctx* c : b.fr) if (c->vox_dense) hashed = false;
    if (hashed) {
        uint32_t maxtiles = 1;
        for (f3ds_ctx* c : b.fr) maxtiles = std::max(maxtiles, (c->n + VT_TILE - 1u) / VT_TILE);
        const int tile_bits = bits_for((uint64_t)maxtiles - 1u);
        rc = for_frames(b, [&](f3ds_ctx* c) { return seg_vox_tiles(c, 3 * maxd, tile_bits); });
        if (rc == F3DS_ERR_UNSUPPORTED) {      // (a grid too deep / a frame too long for this path: nothing was flushed)
            for (f3ds_ctx* c : b.fr) { c->cmds.clear(); c->blob.clear(); c->pend.clear(); c->ops.n = 0; c->ops_grid = 0; c->vox_hashed = false; }
            hashed = false;
        } else if (rc || (rc = flush_sync(b))) return rc;
        if (hashed && g_sw.trace_err) for (f3ds_ctx* c : b.fr) fprintf(stderr, "f3ds: tile path: most voxels in a tile %u, descriptors %u, leaves %u, lists sorted %u\n", c->h_dc->vox_max_tile, c->h_dc->seg_count, c->h_dc->n_voxels, c->h_dc->vox_disorder);
        if (hashed) for (f3ds_ctx* c : b.fr) if (c->h_dc->ev_overflow == 6 || c->h_dc->ev_overflow == 7) {
            if (g_sw.trace_err) fprintf(stderr, "f3ds: tile path refused a frame: %s (descriptors %u, leaves %u)\n", c->h_dc->ev_overflow == 6 ? "a voxel with too many points" : "a tile with too many voxels", c->h_dc->seg_count, c->h_dc->n_voxels);
            c->vox_dense = true; c->vox_dense_wait = c->vox_dense_penalty; c->vox_dense_penalty = std::min(2u * c->vox_dense_penalty, 1024u); hashed = false;
        }
        if (!hashed) {
            if (g_sw.trace_err) fprintf(stderr, "f3ds: voxelisation of %zu frames runs again on the sort path (an unorganised cloud or dense voxels)\n", b.fr.size());
            for (f3ds_ctx* c : b.fr) {
                c->h_dc->error = 0; c->h_dc->ev_overflow = 0; c->vox_hashed = false;
                HIPCHECK(hipMemsetAsync(&c->d_dc->error, 0, sizeof(int), b.st)); HIPCHECK(hipMemsetAsync(&c->d_dc->ev_overflow, 0, sizeof(int), b.st));
                HIPCHECK(hipMemsetAsync(&c->d_dc->n_voxels, 0, sizeof(uint32_t), b.st)); HIPCHECK(hipMemsetAsync(&c->d_dc->seg_count, 0, sizeof(uint32_t), b.st));
            }
        }
    }
    if (!hashed) {
        // (Morton code << idxbits) | point index in one 64-bit word when both fit: the sort then moves 8 bytes per point and pass, not 12
        uint32_t maxn = 1;
        for (f3ds_ctx* c : b.fr) if (c->n > maxn) maxn = c->n;
        int idxbits = bits_for((uint64_t)maxn - 1u);
        if (3 * maxd + 1 + idxbits > 64 || g_sw.sort_pairs) idxbits = -1;
        if ((rc = for_frames(b, [&](f3ds_ctx* c) { return seg_sort(c, 3 * maxd + 1, idxbits); })) || (rc = flush_sync(b))) return rc;
    }
    for (f3ds_ctx* c : b.fr) { if (c->h_dc->error) return trace_err(c->h_dc->error, "voxelisation", c); c->V = c->h_dc->n_voxels; c->res.n_voxels = c->V; }
    if ((rc = drop_dead([](f3ds_ctx* c) { return c->V == 0; }))) return rc;
    stage_mark(b, 1);
    // ---- stage 1 + 2: neighbours, normals, seeds
    if ((rc = for_frames(b, seg_voxels)) || (rc = flush(b))) return rc;
    if ((rc = for_frames(b, seg_normals)) || (rc = flush(b, b.owner->ev[9]))) return rc;      // (the event pair brackets the dispatch alone, not its argument upload)
    stage_mark(b, 10);
    if ((rc = for_frames(b, seg_seed_grid)) || (rc = flush(b))) return rc;
    stage_mark(b, 2);
    if ((rc = flush_sync(b))) return rc;
    int maxsd = 0;
    for (f3ds_ctx* c : b.fr) { if (c->h_dc->error) return c->h_dc->error; if (c->h_dc->sdepth > maxsd) maxsd = c->h_dc->sdepth; }
    if ((rc = for_frames(b, [&](f3ds_ctx* c) { return seg_seed_cells(c, 3 * maxsd, maxsd); })) || (rc = flush_sync(b))) return rc;
    for (f3ds_ctx* c : b.fr) { c->C = c->h_dc->n_cells; c->res.n_seed_cells = c->C; }
    if ((rc = for_frames(b, seg_seeds)) || (rc = flush_sync(b))) return rc;
    for (f3ds_ctx* c : b.fr) { c->S0 = c->h_dc->n_seeds; c->res.n_seeds = c->S0; if (c->S0 >= 0x3FFFFFFFu) return F3DS_ERR_UNSUPPORTED; }      // (labels below 2^30: f3ds_algo.h, a_next_label)
    stage_mark(b, 3);
    // ---- stage 3: sweeps
    if ((rc = for_frames(b, seg_sweeps)) || (rc = flush(b))) return rc;
    stage_mark(b, 4);
    // ---- stage 4: supervoxels, adjacency
    if ((rc = for_frames(b, seg_supervoxels)) || (rc = flush_sync(b))) return rc;
    for (int attempt = 0; attempt < 6; ++attempt) {
        // a frame with more adjacencies than its list holds (tiny supervoxels: 26 neighbours each and more) runs the pass again with four times the room
        bool again = false;
        for (f3ds_ctx* c : b.fr) if (c->h_dc->ev_overflow == 3 && c->edge_mult < (1u << 14)) { c->edge_mult *= 4u; again = true; }
        if (!again) break;
        if (g_sw.trace_err) fprintf(stderr, "f3ds: adjacency pass of %zu frames runs again with more list room\n", b.fr.size());
        for (f3ds_ctx* c : b.fr) {
            c->h_dc->error = 0; c->h_dc->ev_overflow = 0;
            HIPCHECK(hipMemsetAsync(&c->d_dc->error, 0, sizeof(int), b.st)); HIPCHECK(hipMemsetAsync(&c->d_dc->ev_overflow, 0, sizeof(int), b.st));
            HIPCHECK(hipMemsetAsync(&c->d_dc->n_edges, 0, sizeof(uint32_t), b.st)); HIPCHECK(hipMemsetAsync(&c->d_dc->n_alive, 0, sizeof(uint32_t), b.st));
        }
        if ((rc = for_frames(b, seg_supervoxels)) || (rc = flush_sync(b))) return rc;
    }
    uint64_t maxkey = 1;
    for (f3ds_ctx* c : b.fr) {
        if (c->h_dc->error) return trace_err(c->h_dc->error, "supervoxels/adjacency", c);
        if (c->h_dc->r_overflow) return trace_err(F3DS_ERR_UNSUPPORTED, "sweeps (R chain)", c);
        c->E = c->h_dc->n_edges; c->res.n_edges = c->E; c->res.n_supervoxels = c->h_dc->n_alive;
        const uint64_t k = (uint64_t)(c->S0 + 1) * (c->S0 + 1);
        if (k > maxkey) maxkey = k;
    }
    if ((rc = for_frames(b, [&](f3ds_ctx* c) { return seg_edge_sort(c, bits_for(maxkey)); }))) return rc;
    // ---- stage 4c..6
    if ((rc = run_cluster(b, prm, point_labels, index_of, labels_on_device))) return rc;
    HIPCHECK(hipStreamSynchronize(b.st));
    float stage[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // device time of each stage of the whole batch (HIP events on the batch's stream)
    for (int k = 0; k < 7; ++k) { float ms = 0; if (hipEventElapsedTime(&ms, b.owner->ev[k], b.owner->ev[k + 1]) == hipSuccess) stage[k] = ms; }
    // the d_normals launch (inside stage 1): a HIP event pair around that one dispatch on the call's stream (the time it waits at the head of its queue
    // for a compute unit with room for its first workgroup + its execution)
    if (!b.fr.empty()) { float ms = 0; if (hipEventElapsedTime(&ms, b.owner->ev[9], b.owner->ev[10]) == hipSuccess) stage[7] = ms; }
    const float ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (g_sw.host_prof) fprintf(stderr, "batch of %d: %.1f ms total, %.1f waiting for the GPU, %.1f packing + launching, %.1f recording / host logic; stages %.0f %.0f %.0f %.0f %.0f %.0f %.0f ms; %llu scratch allocations so far\n", nctx, ms, g_t_wait, g_t_launch, ms - g_t_wait - g_t_launch, stage[0], stage[1], stage[2], stage[3], stage[4], stage[5], stage[6], g_scratch_allocs.load());
    for (int i = 0; i < nctx; ++i) {
        f3ds_ctx* c = ctxs[i];
        for (int k = 0; k < 8; ++k) c->res.ms_stage[k] = stage[k];
        c->res.ms_total = ms;
        if (results) results[i] = c->res;
    }
    return F3DS_OK;
}

int f3ds_recluster(f3ds_ctx* c, const f3ds_params* prm, uint32_t* point_labels, int labels_on_device, f3ds_result* result) {
    if (!c || !prm) return F3DS_ERR_ARG;
    if (!c->have_frame) return F3DS_ERR_LOGIC;
    const auto t0 = std::chrono::steady_clock::now();
    g_sw.read();
    HIPCHECK(hipSetDevice(c->device));
    Batch b; b.owner = c; b.st = c->stream; b.fr.push_back(c); g_grid_cap = grid_cap_for_batch(1); g_batch_frames = 1;
    c->cmds.clear(); c->blob.clear(); c->pend.clear(); c->ops.n = 0; c->ops_grid = 0;
    c->h_dc->error = 0;
    HIPCHECK(hipMemsetAsync(&c->d_dc->error, 0, sizeof(int), c->stream));
    stage_mark(b, 4);
    std::vector<int> idx{0}; uint32_t* lp[1] = {point_labels};
    int rc = run_cluster(b, prm, lp, idx, labels_on_device);
    if (rc) return rc;
    HIPCHECK(hipStreamSynchronize(b.st));
    for (int k = 4; k < 7; ++k) { float ms = 0; if (hipEventElapsedTime(&ms, c->ev[k], c->ev[k + 1]) == hipSuccess) c->res.ms_stage[k] = ms; }
    c->res.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (result) *result = c->res;
    return F3DS_OK;
}

// Clustering::set_initialstate(segm, adj) + cluster(threshold) on caller-supplied supervoxels (include/f3ds.h).  Host side: the std::map / std::multimap
// orders the reference iterates in (ascending key; adjacency rows by `first`, insertion order inside) become ranks and an edge list; device side: d_sv_user_fill
// builds what d_sv_fill builds for the library's own supervoxels, and stages 4c-6 run unchanged.
int f3ds_cluster_supervoxels(f3ds_ctx* c, const f3ds_supervoxel_set* sv, const uint32_t* pairs, size_t n_pairs, const f3ds_params* prm, uint32_t* region_of_sv,
                             uint32_t* voxel_labels, f3ds_result* result) {
    if (!c || !sv || !prm || (n_pairs && !pairs)) return F3DS_ERR_ARG;
    const uint32_t S = sv->n_supervoxels;
    if (S && (!sv->label || !sv->voxel_offset || !sv->voxel_xyz || !sv->voxel_rgba || !sv->centroid_xyz || !sv->normal)) return F3DS_ERR_ARG;
    if (S >= 0x3FFFFFFFu) return F3DS_ERR_UNSUPPORTED;
    const auto t0 = std::chrono::steady_clock::now();
    g_sw.read();
    HIPCHECK(hipSetDevice(c->device));
    // ---- std::map<uint32_t, Supervoxel::Ptr>: ascending key
    std::vector<uint32_t> row(S + 1u, 0u), lab(S + 1u, 0u), hof(S, 0u);      // row[h], label[h] of internal supervoxel h = rank + 1; hof[row] = h
    {
        std::vector<uint32_t> order(S);
        for (uint32_t i = 0; i < S; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sv->label[a] < sv->label[b]; });
        for (uint32_t r = 0; r < S; ++r) {
            if (r && sv->label[order[r]] == sv->label[order[r - 1]]) return F3DS_ERR_ARG;      // (a map holds a key once)
            row[r + 1] = order[r]; lab[r + 1] = sv->label[order[r]]; hof[order[r]] = r + 1u;
        }
    }
    if (S && sv->voxel_offset[0] != 0u) return F3DS_ERR_ARG;
    for (uint32_t i = 0; i < S; ++i) if (sv->voxel_offset[i + 1] <= sv->voxel_offset[i]) return F3DS_ERR_ARG;      // every supervoxel holds a voxel
    const uint32_t Vt = S ? sv->voxel_offset[S] : 0u;
    if ((uint64_t)Vt + S + 1u > 0x7fffffffull) return F3DS_ERR_UNSUPPORTED;
    // ---- clear_adjacency + adj2weight: rows with first <= second, multimap order
    std::vector<uint32_t> ea, eb;
    {
        std::vector<std::pair<uint32_t, uint32_t>> ed;      // (h_first, h_second)
        std::set<std::pair<uint32_t, uint32_t>> seen;
        for (size_t k = 0; k < n_pairs; ++k) {
            const uint32_t p = pairs[2 * k], q = pairs[2 * k + 1];
            if (p > q) continue;
            uint32_t hh[2];
            for (int w = 0; w < 2; ++w) {
                const uint32_t l = w ? q : p;
                const auto it = std::lower_bound(lab.begin() + 1, lab.end(), l);
                if (it == lab.end() || *it != l) return F3DS_ERR_OUT_OF_RANGE;
                hh[w] = (uint32_t)(it - lab.begin());
            }
            // (defects are reported in pair order, like the map::at / dereference the reference would meet first: an unknown label of pair k wins over a
            // duplicate or self-adjacency of a later pair and the other way round -- the oracle checks in the same order)
            if (hh[0] == hh[1] || !seen.insert({hh[0], hh[1]}).second) return F3DS_ERR_ARG;
            ed.push_back({hh[0], hh[1]});
        }
        std::stable_sort(ed.begin(), ed.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first < b.first; });
        if (ed.size() > 0x7fffffffull) return F3DS_ERR_UNSUPPORTED;
        ea.resize(ed.size()); eb.resize(ed.size());
        for (size_t k = 0; k < ed.size(); ++k) { ea[k] = ed[k].first; eb[k] = ed[k].second; }
    }
    const uint32_t E = (uint32_t)ea.size();
    // ---- frame state
    c->cmds.clear(); c->blob.clear(); c->pend.clear(); c->ops.n = 0; c->ops_grid = 0;
    c->have_frame = false; c->live = true; c->rc = 0; c->refined_itr = -1;
    { const int prc = pregrow_scratch(c); if (prc) return prc; }
    c->prm = *prm; c->n = Vt; c->V = c->C = 0; c->S0 = S; c->E = E; c->d_pts = nullptr;
    memset(&c->res, 0, sizeof c->res);
    c->res.n_voxels = Vt; c->res.n_seeds = S; c->res.n_supervoxels = S; c->res.n_edges = E;
    c->user_mode = true; c->user_label = lab; c->user_row = row;
    if (S == 0) {      // an empty map: cluster() finds an empty weight map and returns (src/clustering.cpp:387)
        c->user_mode = false; c->user_label.clear(); c->user_row.clear(); c->live = false;
        if (result) *result = c->res;
        return F3DS_OK;
    }
    Batch b; b.owner = c; b.st = c->stream; b.fr.push_back(c); g_grid_cap = grid_cap_for_batch(1); g_batch_frames = 1;
    uint32_t *d_src, *d_voff, *d_rgba, *loff, *hcount, *owner, *rcnt0, *ea0, *eb0; float *d_xyz, *d_cent, *d_nrm, *rows, *racc0, *rrec0, *hc; int *row_voxel, *pt_voxel; unsigned char* ralive0;
    ENSURE(c->u_src, uint32_t, S + 1u, d_src); ENSURE(c->u_voff, uint32_t, S + 1u, d_voff); ENSURE(c->u_xyz, float, (size_t)Vt * 3, d_xyz); ENSURE(c->u_rgba, uint32_t, Vt, d_rgba);
    ENSURE(c->u_cent, float, (size_t)S * 3, d_cent); ENSURE(c->u_nrm, float, (size_t)S * 3, d_nrm);
    ENSURE(c->loff, uint32_t, S + 2u, loff); ENSURE(c->hcount, uint32_t, S + 1u, hcount); ENSURE(c->owner0, uint32_t, Vt, owner); ENSURE(c->pt_voxel, int, Vt, pt_voxel);
    ENSURE(c->rows, float, ((size_t)Vt + S + 1) * 12, rows); ENSURE(c->row_voxel, int, (size_t)Vt + S + 1, row_voxel);
    ENSURE(c->racc0, float, (size_t)(S + 1) * 12, racc0); ENSURE(c->rcnt0, uint32_t, S + 1u, rcnt0); ENSURE(c->rrec0, float, (size_t)(S + 1) * 16, rrec0);
    ENSURE(c->ralive0, unsigned char, S + 1u, ralive0); ENSURE(c->hc, float, (size_t)(S + 1) * 12, hc);
    ENSURE(c->ea0, uint32_t, E, ea0); ENSURE(c->eb0, uint32_t, E, eb0);
    std::vector<uint32_t> h_loff(S + 2u, 0u), h_cnt(S + 1u, 0u);
    for (uint32_t h = 1; h <= S; ++h) { h_loff[h] = sv->voxel_offset[row[h]]; h_cnt[h] = sv->voxel_offset[row[h] + 1] - sv->voxel_offset[row[h]]; }
    h_loff[S + 1] = Vt;      // (rows in use: f3ds_get_voxel_cloud)
    const hipStream_t st = b.st;
    HIPCHECK(hipMemcpyAsync(d_src, row.data(), (size_t)(S + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_voff, sv->voxel_offset, (size_t)(S + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_xyz, sv->voxel_xyz, (size_t)Vt * 12, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_rgba, sv->voxel_rgba, (size_t)Vt * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_cent, sv->centroid_xyz, (size_t)S * 12, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_nrm, sv->normal, (size_t)S * 12, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(loff, h_loff.data(), (size_t)(S + 2) * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(hcount, h_cnt.data(), (size_t)(S + 1) * 4, hipMemcpyHostToDevice, st));
    if (E) { HIPCHECK(hipMemcpyAsync(ea0, ea.data(), (size_t)E * 4, hipMemcpyHostToDevice, st)); HIPCHECK(hipMemcpyAsync(eb0, eb.data(), (size_t)E * 4, hipMemcpyHostToDevice, st)); }
    stage_mark(b, 4);
    rec<d_init_counters>(c, 1u, 0u, c->d_dc);
    rec_fill(c, ralive0, 0u, S + 1u);
    rec_fill(c, rcnt0, 0u, (size_t)(S + 1) * 4);
    rec_fill(c, racc0, 0u, 48); rec_fill(c, rrec0, 0u, 64); rec_fill(c, hc, 0u, 48);
    rec<d_iota>(c, grid_for(Vt, 256), 0u, (uint32_t*)pt_voxel, Vt);
    rec<d_sv_user_fill>(c, S, 0u, S, (const uint32_t*)d_src, (const uint32_t*)d_voff, (const float*)d_xyz, (const uint32_t*)d_rgba, (const float*)d_cent, (const float*)d_nrm, rows, row_voxel,
                        owner, racc0, rcnt0, rrec0, ralive0, hc, c->d_dc);
    int rc = flush(b);
    if (rc) return rc;
    uint32_t* lp[1] = {voxel_labels}; std::vector<int> idx{0};
    if ((rc = run_cluster(b, prm, lp, idx, 0))) return rc;
    HIPCHECK(hipStreamSynchronize(b.st));
    for (int k = 4; k < 7; ++k) { float ms = 0; if (hipEventElapsedTime(&ms, c->ev[k], c->ev[k + 1]) == hipSuccess) c->res.ms_stage[k] = ms; }
    if (region_of_sv) {
        std::vector<uint32_t> root(S + 1u);
        HIPCHECK(hipMemcpy(root.data(), c->root.p, (size_t)(S + 1) * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < S; ++i) region_of_sv[i] = lab[root[hof[i]]];
    }
    c->live = false;
    c->res.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (result) *result = c->res;
    return F3DS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// accessors (not on the hot path): copy device state to the host and repack
// ------------------------------------------------------------------------------------------------
namespace {
template <class T>
int fetch(f3ds_ctx* c, const Buf& b, size_t count, std::vector<T>& out) {
    out.resize(count);
    if (count) HIPCHECK(hipMemcpy(out.data(), b.p, count * sizeof(T), hipMemcpyDeviceToHost));
    return F3DS_OK;
}
// label of internal supervoxel h as the caller knows it: h itself after f3ds_segment (helper labels 1..S0), the caller's own after f3ds_cluster_supervoxels
inline uint32_t label_out(const f3ds_ctx* c, uint32_t h) { return c->user_mode && h < c->user_label.size() ? c->user_label[h] : h; }
// the leaves of every alive region, in voxels_ concatenation order, as (first payload row, rows)
int region_leaves(f3ds_ctx* c, std::vector<unsigned char>& ralive, std::vector<std::vector<uint2>>& leaves) {
    const uint32_t S0 = c->S0;
    std::vector<uint32_t> rhead, lnext, loff, llen, rstart, rnleaf; std::vector<uint2> pool;
    int rc;
    if ((rc = fetch(c, c->ralive, S0 + 1, ralive)) || (rc = fetch(c, c->rhead, S0 + 1, rhead)) || (rc = fetch(c, c->lnext, S0 + 1, lnext)) ||
        (rc = fetch(c, c->loff, S0 + 2, loff)) || (rc = fetch(c, c->hcount, S0 + 1, llen)))
        return rc;
    if (c->merge_in_lds && ((rc = fetch(c, c->pool, c->pool.cap / 8, pool)) || (rc = fetch(c, c->rstart, S0 + 1, rstart)) || (rc = fetch(c, c->rnleaf, S0 + 1, rnleaf)))) return rc;
    leaves.assign(S0 + 1, {});
    for (uint32_t h = 1; h <= S0; ++h) {
        if (!ralive[h]) continue;
        if (c->merge_in_lds) leaves[h].assign(pool.begin() + rstart[h], pool.begin() + rstart[h] + rnleaf[h]);
        else for (uint32_t leaf = rhead[h]; leaf; leaf = lnext[leaf]) leaves[h].push_back(make_uint2(loff[leaf], llen[leaf]));
    }
    return F3DS_OK;
}
// payload rows in use: the library's own supervoxels are packed (loff[S0 + 1] = total); a caller's keep the caller's layout
inline size_t rows_in_use(const f3ds_ctx* c, const std::vector<uint32_t>& loff) { return c->user_mode ? c->n : loff[c->S0 + 1]; }
}  // namespace

// get_currentstate().first: the merged regions in ascending key (include/f3ds.h)
extern "C" int f3ds_get_regions(f3ds_ctx* c, uint32_t* label, uint32_t* n_voxels, float* centroid_xyz, float* normal, float* mean_rgb, size_t cap, size_t* n_out) {
    if (!c) return F3DS_ERR_ARG;
    if (!c->have_frame) return F3DS_ERR_LOGIC;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    const uint32_t S0 = c->S0;
    std::vector<unsigned char> ralive; std::vector<uint32_t> rcnt; std::vector<float> rrec;
    int rc;
    if ((rc = fetch(c, c->ralive, S0 + 1, ralive)) || (rc = fetch(c, c->rcnt, S0 + 1, rcnt)) || (rc = fetch(c, c->rrec, (size_t)(S0 + 1) * 16, rrec))) return rc;
    size_t k = 0;
    for (uint32_t h = 1; h <= S0; ++h) {
        if (!ralive[h]) continue;
        if (k < cap) {
            const float* r = &rrec[(size_t)h * 16];
            if (label) label[k] = label_out(c, h);
            if (n_voxels) n_voxels[k] = rcnt[h];
            if (centroid_xyz) { centroid_xyz[3 * k] = r[0]; centroid_xyz[3 * k + 1] = r[1]; centroid_xyz[3 * k + 2] = r[2]; }
            if (normal) { normal[3 * k] = r[3]; normal[3 * k + 1] = r[4]; normal[3 * k + 2] = r[5]; }
            if (mean_rgb) { mean_rgb[3 * k] = r[6]; mean_rgb[3 * k + 1] = r[7]; mean_rgb[3 * k + 2] = r[8]; }
        }
        k++;
    }
    if (n_out) *n_out = k;
    return (k > cap && (label || n_voxels || centroid_xyz || normal || mean_rgb)) ? F3DS_ERR_CAPACITY : F3DS_OK;
}

extern "C" int f3ds_get_region_voxels(f3ds_ctx* c, float* xyz, uint32_t* rgba, uint32_t* voxel_index, size_t cap, size_t* n_out) {
    if (!c) return F3DS_ERR_ARG;
    if (!c->have_frame) return F3DS_ERR_LOGIC;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    std::vector<unsigned char> ralive; std::vector<std::vector<uint2>> leaves; std::vector<float> rows; std::vector<int> rv; std::vector<uint32_t> loff;
    int rc;
    if ((rc = region_leaves(c, ralive, leaves)) || (rc = fetch(c, c->loff, c->S0 + 2, loff))) return rc;
    const size_t nrows = rows_in_use(c, loff);
    if ((rc = fetch(c, c->rows, nrows * 12, rows)) || (rc = fetch(c, c->row_voxel, nrows, rv))) return rc;
    size_t k = 0;
    for (uint32_t h = 1; h <= c->S0; ++h)
        for (const uint2& leaf : leaves[h])
            for (uint32_t j = 0; j < leaf.y; ++j) {
                if (k < cap) {
                    const float* r = &rows[(size_t)(leaf.x + j) * 12];
                    if (xyz) { xyz[3 * k] = r[6]; xyz[3 * k + 1] = r[7]; xyz[3 * k + 2] = r[8]; }
                    if (rgba) rgba[k] = (uint32_t)r[9] << 16 | (uint32_t)r[10] << 8 | (uint32_t)r[11];
                    if (voxel_index) voxel_index[k] = (uint32_t)rv[leaf.x + j];
                }
                k++;
            }
    if (n_out) *n_out = k;
    return (k > cap && (xyz || rgba || voxel_index)) ? F3DS_ERR_CAPACITY : F3DS_OK;
}

extern "C" int f3ds_get_voxel_cloud(f3ds_ctx* c, float* xyz, uint32_t* label, uint32_t* rgba, size_t cap, size_t* n_out) {
    if (!c) return F3DS_ERR_ARG;
    if (!c->have_frame) return F3DS_ERR_LOGIC;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    const uint32_t S0 = c->S0;
    std::vector<unsigned char> ralive; std::vector<std::vector<uint2>> all_leaves; std::vector<uint32_t> loff; std::vector<float> rows;
    int rc;
    if ((rc = region_leaves(c, ralive, all_leaves)) || (rc = fetch(c, c->loff, S0 + 2, loff))) return rc;
    if ((rc = fetch(c, c->rows, rows_in_use(c, loff) * 12, rows))) return rc;
    size_t k = 0; uint32_t cur = 0;
    for (uint32_t h = 1; h <= S0; ++h) {
        if (!ralive[h]) continue;
        const std::vector<uint2>& leaves = all_leaves[h];      // the region's leaves (first payload row, rows) in voxels_ concatenation order
        for (const uint2& leaf : leaves)
            for (uint32_t j = 0; j < leaf.y; ++j) {
                if (k < cap) {
                    const float* r = &rows[(size_t)(leaf.x + j) * 12];
                    if (xyz) { xyz[3 * k] = r[6]; xyz[3 * k + 1] = r[7]; xyz[3 * k + 2] = r[8]; }
                    if (label) label[k] = cur;
                    if (rgba) rgba[k] = f3ds_glasbey_256[cur % 256u];
                }
                k++;
            }
        cur++;
    }
    if (n_out) *n_out = k;
    return (k > cap && (xyz || label || rgba)) ? F3DS_ERR_CAPACITY : F3DS_OK;
}

extern "C" int f3ds_get_voxel_centroid_cloud(f3ds_ctx* c, float* xyz, uint32_t* rgba, uint32_t* sv_label, size_t cap, size_t* n_out) {
    if (!c) return F3DS_ERR_ARG;
    if (!c->have_frame || c->user_mode) return F3DS_ERR_LOGIC;      // (caller-supplied supervoxels have no voxel grid)
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    const uint32_t V = c->V;
    if (n_out) *n_out = V;
    if (!xyz && !rgba && !sv_label) return F3DS_OK;
    if (cap < V) return F3DS_ERR_CAPACITY;
    std::vector<float> f; std::vector<uint32_t> o;
    int rc;
    if ((rc = fetch(c, c->vf, (size_t)V * 12, f)) || (rc = fetch(c, c->owner0, V, o))) return rc;
    for (uint32_t v = 0; v < V; ++v) {
        const float* r = &f[(size_t)v * 12];
        if (xyz) { xyz[3 * v] = r[0]; xyz[3 * v + 1] = r[1]; xyz[3 * v + 2] = r[2]; }
        if (rgba) rgba[v] = ((uint32_t)r[3] & 255u) << 16 | ((uint32_t)r[4] & 255u) << 8 | ((uint32_t)r[5] & 255u);
        if (sv_label) sv_label[v] = o[v];
    }
    return F3DS_OK;
}

extern "C" int f3ds_get_supervoxels(f3ds_ctx* c, uint32_t* label, float* xyz, float* rgb, float* normal, uint32_t* n_voxels, size_t cap, size_t* n_out);

// SupervoxelClustering::refineSupervoxels(num_itr, refined) [PCL-recall], called at src/supervoxel_clustering.cpp:371: on copies of
// the sweep state (main() goes on clustering the unrefined supervoxels), num_itr x { refineNormals on every helper's leaves;
// reseedSupervoxels; expandSupervoxels(max_depth) }.
extern "C" int f3ds_refine_supervoxels(f3ds_ctx* c, int num_itr) {
    if (!c || num_itr < 0) return F3DS_ERR_ARG;
    if (!c->have_frame || c->user_mode) return F3DS_ERR_LOGIC;
    g_sw.read();
    HIPCHECK(hipSetDevice(c->device));
    Batch b; b.owner = c; b.st = c->stream; b.fr.push_back(c); g_grid_cap = grid_cap_for_batch(1); g_batch_frames = 1;
    c->cmds.clear(); c->blob.clear(); c->pend.clear(); c->ops.n = 0; c->ops_grid = 0;
    c->refined_itr = -1;
    const uint32_t V = c->V, S0 = c->S0;
    float *r_vf, *r_hc; uint32_t *r_owner, *r_hcount, *L; int *r_gvox, *seed; unsigned char* r_gact;
    ENSURE(c->r_vf, float, (size_t)V * 12, r_vf); ENSURE(c->r_owner, uint32_t, V, r_owner); ENSURE(c->r_hc, float, (size_t)(S0 + 1) * 12, r_hc);
    ENSURE(c->r_hcount, uint32_t, S0 + 1, r_hcount); ENSURE(c->r_gvox, int, S0 + 1, r_gvox); ENSURE(c->r_gact, unsigned char, S0 + 1, r_gact);
    ENSURE(c->r_seed, int, S0 + 1, seed); ENSURE(c->r_L, uint32_t, V, L);
    rec_copy(c, r_vf, c->vf.p, (size_t)V * 48); rec_copy(c, r_owner, c->owner0.p, (size_t)V * 4); rec_copy(c, r_hc, c->hc.p, (size_t)(S0 + 1) * 48);
    rec_copy(c, r_hcount, c->hcount.p, (size_t)(S0 + 1) * 4); rec_copy(c, r_gvox, c->ghost_vox.p, (size_t)(S0 + 1) * 4); rec_copy(c, r_gact, c->ghost_active.p, S0 + 1);
    SweepBufs sb{&c->r_owner, &c->r_dist, &c->r_hc, &c->r_hcount, &c->r_hlo, &c->r_hhi, &c->r_gvox, &c->r_gact, &c->r_gdone, &c->r_ghead, &c->r_gnext,
                 &c->r_tl, &c->r_tcnt, r_vf, seed, true};
    int rc;
    for (int it = 0; it < num_itr; ++it) {
        rec_copy(c, L, r_owner, (size_t)V * 4);
        rec<d_refine_ghost_L>(c, grid_for(S0, 256), 0u, S0, (const int*)r_gvox, (const unsigned char*)r_gact, L);
        rec<d_refine_normals>(c, grid_for(V, 256), 0u, r_vf, (const int*)c->nbr.p, (const uint32_t*)r_owner, (const uint32_t*)L, V);
        rec<d_reseed>(c, S0 ? S0 : 1u, 0u, (const float*)r_vf, V, S0, (const uint32_t*)r_hcount, (const float*)r_hc, seed);
        if ((rc = seg_sweeps_on(c, sb)) || (rc = flush_sync(b))) return rc;
        if (c->h_dc->error) return trace_err(c->h_dc->error, "refine", c);
    }
    if ((rc = flush_sync(b))) return rc;
    c->refined_itr = num_itr;
    return F3DS_OK;
}

// the refined state: per voxel (leaf order) its supervoxel label (0 = none) and normal
extern "C" int f3ds_get_refined_voxels(f3ds_ctx* c, uint32_t* sv_label, float* normal, size_t cap, size_t* n_out) {
    if (!c) return F3DS_ERR_ARG;
    if (!c->have_frame || c->refined_itr < 0) return F3DS_ERR_LOGIC;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    const uint32_t V = c->V;
    if (n_out) *n_out = V;
    if (!sv_label && !normal) return F3DS_OK;
    if (cap < V) return F3DS_ERR_CAPACITY;
    std::vector<float> f; std::vector<uint32_t> o;
    int rc;
    if ((rc = fetch(c, c->r_vf, (size_t)V * 12, f)) || (rc = fetch(c, c->r_owner, V, o))) return rc;
    for (uint32_t v = 0; v < V; ++v) {
        if (sv_label) sv_label[v] = o[v];
        if (normal) { normal[3 * v] = f[(size_t)v * 12 + 6]; normal[3 * v + 1] = f[(size_t)v * 12 + 7]; normal[3 * v + 2] = f[(size_t)v * 12 + 8]; }
    }
    return F3DS_OK;
}

namespace {
int supervoxels_of(f3ds_ctx* c, const Buf& hcount, const Buf& hcent, uint32_t* label, float* xyz, float* rgb, float* normal, uint32_t* n_voxels, size_t cap, size_t* n_out) {
    const uint32_t S0 = c->S0;
    std::vector<uint32_t> cnt; std::vector<float> hc;
    int rc;
    if ((rc = fetch(c, hcount, S0 + 1, cnt)) || (rc = fetch(c, hcent, (size_t)(S0 + 1) * 12, hc))) return rc;
    size_t k = 0;
    for (uint32_t h = 1; h <= S0; ++h) {
        if (!cnt[h]) continue;
        if (k < cap) {
            const float* r = &hc[(size_t)h * 12];
            if (label) label[k] = label_out(c, h);
            if (xyz) { xyz[3 * k] = r[0]; xyz[3 * k + 1] = r[1]; xyz[3 * k + 2] = r[2]; }
            if (rgb) { rgb[3 * k] = r[3]; rgb[3 * k + 1] = r[4]; rgb[3 * k + 2] = r[5]; }
            if (normal) { normal[3 * k] = r[6]; normal[3 * k + 1] = r[7]; normal[3 * k + 2] = r[8]; }
            if (n_voxels) n_voxels[k] = cnt[h];
        }
        k++;
    }
    if (n_out) *n_out = k;
    return (k > cap && (label || xyz || rgb || normal || n_voxels)) ? F3DS_ERR_CAPACITY : F3DS_OK;
}
}  // namespace

// the refined supervoxel map (refined_supervoxel_clusters): same layout as f3ds_get_supervoxels
extern "C" int f3ds_get_refined_supervoxels(f3ds_ctx* c, uint32_t* label, float* xyz, float* rgb, float* normal, uint32_t* n_voxels, size_t cap, size_t* n_out) {
    if (!c) return F3DS_ERR_ARG;
    if (!c->have_frame || c->refined_itr < 0) return F3DS_ERR_LOGIC;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return supervoxels_of(c, c->r_hcount, c->r_hc, label, xyz, rgb, normal, n_voxels, cap, n_out);
}

extern "C" int f3ds_get_supervoxels(f3ds_ctx* c, uint32_t* label, float* xyz, float* rgb, float* normal, uint32_t* n_voxels, size_t cap, size_t* n_out) {
    if (!c) return F3DS_ERR_ARG;
    if (!c->have_frame) return F3DS_ERR_LOGIC;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return supervoxels_of(c, c->hcount, c->hc, label, xyz, rgb, normal, n_voxels, cap, n_out);
}

extern "C" int f3ds_get_supervoxel_adjacency(f3ds_ctx* c, uint32_t* pairs, size_t cap_pairs, size_t* n_out) {
    if (!c) return F3DS_ERR_ARG;
    if (!c->have_frame) return F3DS_ERR_LOGIC;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    const uint32_t E = c->E;
    if (n_out) *n_out = E;
    if (!pairs) return F3DS_OK;
    if (cap_pairs < E) return F3DS_ERR_CAPACITY;
    std::vector<uint32_t> a, b;
    int rc;
    if ((rc = fetch(c, c->ea0, E, a)) || (rc = fetch(c, c->eb0, E, b))) return rc;
    for (uint32_t e = 0; e < E; ++e) { pairs[2 * e] = label_out(c, a[e]); pairs[2 * e + 1] = label_out(c, b[e]); }
    return F3DS_OK;
}

// Clustering::get_currentstate().second after cluster(): the adjacency of the merged regions, as pairs (a < b) of the
// surviving supervoxel labels, sorted (src/supervoxel_clustering.cpp:444,465: what visualize() draws as the graph).  Every
// initial adjacency maps to the labels its two supervoxels ended in; merge() keeps one edge per pair (contains(), clustering.cpp:408-436).
extern "C" int f3ds_get_region_adjacency(f3ds_ctx* c, uint32_t* pairs, size_t cap_pairs, size_t* n_out) {
    if (!c) return F3DS_ERR_ARG;
    if (!c->have_frame) return F3DS_ERR_LOGIC;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    std::vector<uint32_t> a, b, root;
    int rc;
    if ((rc = fetch(c, c->ea0, c->E, a)) || (rc = fetch(c, c->eb0, c->E, b)) || (rc = fetch(c, c->root, c->S0 + 1, root))) return rc;
    std::vector<uint64_t> keys;
    for (uint32_t e = 0; e < c->E; ++e) {
        const uint32_t p = root[a[e]], q = root[b[e]];
        if (p != q) keys.push_back(((uint64_t)(p < q ? p : q) << 32) | (p < q ? q : p));
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    if (n_out) *n_out = keys.size();
    if (!pairs) return F3DS_OK;
    if (cap_pairs < keys.size()) return F3DS_ERR_CAPACITY;
    for (size_t k = 0; k < keys.size(); ++k) { pairs[2 * k] = label_out(c, (uint32_t)(keys[k] >> 32)); pairs[2 * k + 1] = label_out(c, (uint32_t)keys[k]); }
    return F3DS_OK;
}

extern "C" int f3ds_merge_layout_info(uint32_t n_edges, int waves, int keys_in_lds, uint32_t out[4]) {
    if (!out || (waves != 4 && waves != 8) || (keys_in_lds != 0 && keys_in_lds != 2)) return F3DS_ERR_ARG;
    const MergeIlLayout L = merge_il_offsets(n_edges, waves, keys_in_lds);
    out[0] = L.total; out[1] = L.sp_rows; out[2] = L.Ecap;
    out[3] = (L.total <= MC_LDS_LIMIT && (uint64_t)n_edges * (keys_in_lds == 2 ? 10u : 2u) <= (1u << 22)) ? 1u : 0u;
    return F3DS_OK;
}
extern "C" int f3ds_get_debug(f3ds_ctx* c, int what, void* dst, size_t cap_bytes, size_t* bytes_out) {
    if (!c) return F3DS_ERR_ARG;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    const uint32_t V = c->V, S0 = c->S0, E = c->E, n = c->n;
    std::vector<uint8_t> buf;
    auto put = [&](const void* p, size_t nb) { const uint8_t* b = (const uint8_t*)p; buf.insert(buf.end(), b, b + nb); };
    int rc = F3DS_OK;
    std::vector<float> f; std::vector<uint32_t> u, u2, u3; std::vector<int> iv; std::vector<unsigned char> uc;
    if (what != F3DS_DBG_GRID && !c->have_frame) return F3DS_ERR_LOGIC;
    switch (what) {
        case F3DS_DBG_GRID: { HIPCHECK(hipMemcpy(c->h_grid, c->d_grid, sizeof(GridInfo), hipMemcpyDeviceToHost)); double g[5] = {c->h_grid->min[0], c->h_grid->min[1], c->h_grid->min[2], c->h_grid->res, (double)c->h_grid->depth}; put(g, sizeof g); break; }
        case F3DS_DBG_TILE_LIST_LEN: if ((rc = fetch(c, c->tile_n1, (size_t)(V + NT_TILE - 1) / NT_TILE, u))) return rc; put(u.data(), u.size() * 4); break;
        case F3DS_DBG_STAGE0_PATH: { const uint32_t w = c->vox_hashed ? 1u : 0u; put(&w, 4); break; }
        case F3DS_DBG_SWEEP_STATS: {
            DevCounters dcs;
            HIPCHECK(hipMemcpy(&dcs, c->d_dc, sizeof dcs, hipMemcpyDeviceToHost));
            put(dcs.sweep_stats, 16);
#ifdef F3DS_CEN_STATS
            fprintf(stderr, "d_centroid: quads %u, live rows %u, to the wave path: list %u, leaves %u; per quad ndmax %.2f, max leaves %.1f; per row nd %.2f, leaves %.1f\n", dcs.cen_stats[0], dcs.cen_stats[1], dcs.cen_stats[2],
                    dcs.cen_stats[3], dcs.cen_stats[4] / (double)dcs.cen_stats[0], dcs.cen_stats[7] / (double)dcs.cen_stats[0], dcs.cen_stats[5] / (double)dcs.cen_stats[1], dcs.cen_stats[6] / (double)dcs.cen_stats[1]);
#endif
            break;
        }
        case F3DS_DBG_MERGE_LAYOUT: { const uint32_t w[2] = {c->merge_kind == MK_GLOBAL ? 0u : (uint32_t)mk_waves(c->merge_kind), c->merge_kind == MK_GLOBAL ? 0u : (uint32_t)mk_res(c->merge_kind)}; put(w, 8); break; }
        case F3DS_DBG_VOXEL_KEYS: if ((rc = fetch(c, c->vkey, (size_t)V * 3, u))) return rc; put(u.data(), u.size() * 4); break;
        case F3DS_DBG_VOXEL_COUNT: if ((rc = fetch(c, c->vcount, V, u))) return rc; put(u.data(), u.size() * 4); break;
        case F3DS_DBG_VOXEL_XYZ: case F3DS_DBG_VOXEL_RGB: case F3DS_DBG_VOXEL_NORMAL:
            if ((rc = fetch(c, c->vf, (size_t)V * 12, f))) return rc;
            for (uint32_t v = 0; v < V; ++v) {
                if (what == F3DS_DBG_VOXEL_XYZ) put(&f[(size_t)v * 12], 12);
                else if (what == F3DS_DBG_VOXEL_RGB) put(&f[(size_t)v * 12 + 3], 12);
                else { float n4[4] = {f[(size_t)v * 12 + 6], f[(size_t)v * 12 + 7], f[(size_t)v * 12 + 8], 0.0f}; put(n4, 16); }
            }
            break;
        case F3DS_DBG_VOXEL_NEIGHBORS: if ((rc = fetch(c, c->nbr, (size_t)V * 27, iv))) return rc; put(iv.data(), iv.size() * 4); break;
        case F3DS_DBG_POINT_VOXEL: if ((rc = fetch(c, c->pt_voxel, n, iv))) return rc; put(iv.data(), iv.size() * 4); break;
        case F3DS_DBG_SEED_ORIG: if ((rc = fetch(c, c->seed_orig, c->C, iv))) return rc; put(iv.data(), iv.size() * 4); break;
        case F3DS_DBG_SEED_KEPT: if ((rc = fetch(c, c->seed_kept, S0, iv))) return rc; put(iv.data(), iv.size() * 4); break;
        case F3DS_DBG_VOXEL_SVLABEL: if ((rc = fetch(c, c->owner0, V, u))) return rc; put(u.data(), u.size() * 4); break;
        case F3DS_DBG_VOXEL_DIST: if ((rc = fetch(c, c->dist0, V, f))) return rc; put(f.data(), f.size() * 4); break;
        case F3DS_DBG_SV_LABELS: case F3DS_DBG_SV_CENTROID: case F3DS_DBG_SV_REGION:
            if ((rc = fetch(c, c->hcount, S0 + 1, u)) || (rc = fetch(c, c->hc, (size_t)(S0 + 1) * 12, f)) || (rc = fetch(c, c->root, S0 + 1, u2))) return rc;
            for (uint32_t h = 1; h <= S0; ++h) {
                if (!u[h]) continue;
                const uint32_t lh = label_out(c, h), lr = label_out(c, u2[h]);
                if (what == F3DS_DBG_SV_LABELS) put(&lh, 4);
                else if (what == F3DS_DBG_SV_REGION) put(&lr, 4);
                else { float r[10]; for (int k = 0; k < 9; ++k) r[k] = f[(size_t)h * 12 + k]; r[9] = 0.0f; put(r, 40); }
            }
            break;
        case F3DS_DBG_EDGES:
            if ((rc = fetch(c, c->ea0, E, u)) || (rc = fetch(c, c->eb0, E, u2))) return rc;
            for (uint32_t e = 0; e < E; ++e) { const uint32_t la = label_out(c, u[e]), lb = label_out(c, u2[e]); put(&la, 4); put(&lb, 4); }
            break;
        case F3DS_DBG_EDGE_DELTAS: if ((rc = fetch(c, c->deltas, (size_t)E * 2, f))) return rc; put(f.data(), f.size() * 4); break;
        case F3DS_DBG_EDGE_WEIGHTS: {
            // initial weights = the epoch-0 history events, decoded back from their keys would lose NaN payloads:
            // recompute from deltas on the host with the same arithmetic
            if ((rc = fetch(c, c->deltas, (size_t)E * 2, f))) return rc;
            MergeParams mp; mp.color_metric = c->prm.color_metric; mp.geom_metric = c->prm.geom_metric; mp.merging = c->prm.merging;
            mp.lambda = c->res.lambda; mp.bins = (c->prm.merging == F3DS_EQUALIZATION && c->prm.bins != 0) ? (short)c->prm.bins : 500;
            std::vector<float> cdf;
            if (c->prm.merging == F3DS_EQUALIZATION) { if ((rc = fetch(c, c->cdf, (size_t)2 * mp.bins, cdf))) return rc; mp.cdf_c = cdf.data(); mp.cdf_g = cdf.data() + mp.bins; }
            else { mp.cdf_c = mp.cdf_g = nullptr; }
            for (uint32_t e = 0; e < E; ++e) { int err = 0; float w = a_tc(mp, f[e * 2], &err) + a_tg(mp, f[e * 2 + 1], &err); put(&w, 4); }
            break;
        }
        case F3DS_DBG_MERGES:
            if ((rc = fetch(c, c->merges, (size_t)c->res.n_merges * 3, u))) return rc;
            if (c->user_mode) for (size_t k = 0; k + 2 < u.size(); k += 3) { u[k] = label_out(c, u[k]); u[k + 1] = label_out(c, u[k + 1]); }
            put(u.data(), u.size() * 4); break;
        case F3DS_DBG_VOXEL_REGION:
            if ((rc = fetch(c, c->owner0, V, u)) || (rc = fetch(c, c->root, S0 + 1, u2)) || (rc = fetch(c, c->rincl, S0 + 1, u3))) return rc;
            for (uint32_t v = 0; v < V; ++v) { uint32_t l = u[v] ? u3[u2[u[v]]] - 1u : F3DS_NO_LABEL; put(&l, 4); }
            break;
        default: return F3DS_ERR_ARG;
    }
    if (bytes_out) *bytes_out = buf.size();
    if (dst) {
        if (buf.size() > cap_bytes) return F3DS_ERR_CAPACITY;
        if (!buf.empty()) memcpy(dst, buf.data(), buf.size());
    }
    return F3DS_OK;
}

// ------------------------------------------------------------------------------------------------
// ground-truth evaluation and the automatic threshold (reference main():387-447, Clustering::all_thresh
// / best_thresh clustering.cpp:691-774, Testing src/testing.cpp).  Not on the per-frame hot path.
// ------------------------------------------------------------------------------------------------
namespace {
// truth label of every voxel: label colours averaged per voxel on the GPU, distinct colours numbered
// in first-appearance (leaf) order on the host (color2label, clustering.cpp:823-846)
int eval_truth(f3ds_ctx* c, const uint32_t* truth_point_labels) {
    const uint32_t n = c->n, V = c->V;
    uint32_t *lut, *tp, *tsum, *tcol, *tlab;
    ENSURE(c->glut, uint32_t, 256, lut); ENSURE(c->truth_pts, uint32_t, n, tp); ENSURE(c->tsum, uint32_t, (size_t)V * 3, tsum);
    ENSURE(c->tcol, uint32_t, V, tcol); ENSURE(c->tlab, uint32_t, V, tlab);
    HIPCHECK(hipMemcpyAsync(lut, f3ds_glasbey_256, 1024, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(tp, truth_point_labels, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemsetAsync(tsum, 0, (size_t)V * 12, c->stream));
    Batch b; b.owner = c; b.st = c->stream; b.fr.push_back(c); g_grid_cap = grid_cap_for_batch(1); g_batch_frames = 1;
    c->cmds.clear(); c->blob.clear(); c->pend.clear(); c->ops.n = 0; c->ops_grid = 0;
    rec<d_truth_accum>(c, grid_for(n, 256), 0u, n, (const int*)c->pt_voxel.p, (const uint32_t*)tp, (const uint32_t*)lut, tsum);
    rec<d_truth_color>(c, grid_for(V, 256), 0u, V, (const uint32_t*)tsum, (const uint32_t*)c->vcount.p, tcol);
    int rc = flush_sync(b);
    if (rc) return rc;
    std::vector<uint32_t> col;
    if ((rc = fetch(c, c->tcol, V, col))) return rc;
    std::map<uint32_t, uint32_t> ids;
    c->tsize.clear();
    for (uint32_t v = 0; v < V; ++v) {
        auto it = ids.find(col[v]);
        uint32_t l;
        if (it == ids.end()) { l = (uint32_t)c->tsize.size(); ids.insert({col[v], l}); c->tsize.push_back(0); } else l = it->second;
        col[v] = l; c->tsize[l]++;
    }
    HIPCHECK(hipMemcpy(tlab, col.data(), (size_t)V * 4, hipMemcpyHostToDevice));
    return F3DS_OK;
}
// scores of the segmentation described by (root, incl): incl[root[h]]-1 is the segment of supervoxel h
int eval_scores(f3ds_ctx* c, const uint32_t* d_root, const uint32_t* d_incl, uint32_t K, f3ds_performance* out) {
    const uint32_t V = c->V, M = (uint32_t)c->tsize.size();
    if (K == 0 || M == 0) return F3DS_ERR_ARG;                         // std::invalid_argument, testing.cpp:414,431
    if ((uint64_t)K * M > (1ull << 26)) return F3DS_ERR_UNSUPPORTED;
    uint32_t *tab, *ssz;
    ENSURE(c->ctab, uint32_t, (size_t)K * M, tab); ENSURE(c->csize, uint32_t, K, ssz);
    HIPCHECK(hipMemsetAsync(tab, 0, (size_t)K * M * 4, c->stream));
    HIPCHECK(hipMemsetAsync(ssz, 0, (size_t)K * 4, c->stream));
    Batch b; b.owner = c; b.st = c->stream; b.fr.push_back(c); g_grid_cap = grid_cap_for_batch(1); g_batch_frames = 1;
    c->cmds.clear(); c->blob.clear(); c->pend.clear(); c->ops.n = 0; c->ops_grid = 0;
    rec<d_contingency>(c, grid_for(V, 256), 0u, V, M, (const uint32_t*)c->owner0.p, d_root, d_incl, (const uint32_t*)c->tlab.p, tab, ssz);
    rec<d_contingency_ghost>(c, grid_for(c->S0, 256), 0u, c->S0, M, (const int*)c->ghost_vox.p, (const unsigned char*)c->ghost_active.p,
                             (const uint32_t*)c->owner0.p, d_root, d_incl, (const uint32_t*)c->tlab.p, tab, ssz);
    int rc = flush_sync(b);
    if (rc) return rc;
    std::vector<uint32_t> table, ssize;
    if ((rc = fetch(c, c->ctab, (size_t)K * M, table)) || (rc = fetch(c, c->csize, K, ssize))) return rc;
    *out = f3ds_scores_from_table(table, ssize, c->tsize, V);
    return F3DS_OK;
}
}  // namespace

extern "C" int f3ds_evaluate(f3ds_ctx* c, const uint32_t* truth_point_labels, f3ds_performance* out) {
    if (!c || !truth_point_labels || !out) return F3DS_ERR_ARG;
    if (!c->have_frame || c->user_mode) return F3DS_ERR_LOGIC;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    int rc = eval_truth(c, truth_point_labels);
    if (rc) return rc;
    return eval_scores(c, (const uint32_t*)c->root.p, (const uint32_t*)c->rincl.p, c->res.n_regions, out);
}

extern "C" int f3ds_auto_threshold(f3ds_ctx* c, const f3ds_params* prm, const uint32_t* truth_point_labels, float start, float end, float step,
                                   float* thresholds, f3ds_performance* scores, size_t cap, size_t* n_out, float* best_threshold,
                                   f3ds_performance* best_score, uint32_t* point_labels, int labels_on_device, f3ds_result* result) {
    if (!c || !prm || !truth_point_labels) return F3DS_ERR_ARG;
    if (!c->have_frame || c->user_mode) return F3DS_ERR_LOGIC;
    if (start < 0 || start > 1 || end < 0 || end > 1 || !(step > 0) || step > 1) return F3DS_ERR_OUT_OF_RANGE;   // std::out_of_range, clustering.cpp:694-698 (step 0 never ends there)
    if (start > end) { float t = start; start = end; end = t; }
    std::vector<float> ts{start};
    for (float t = start + step; t <= end; t += step) ts.push_back(t);
    // Clustering::cluster(state, t) continues one merge sequence, so every threshold's segmentation is a
    // prefix of the merges done at the largest one: merge once, then replay the log on the host
    f3ds_params p = *prm;
    p.threshold = ts.back();
    int rc = f3ds_recluster(c, &p, nullptr, 0, nullptr);
    if (rc) return rc;
    const uint32_t S0 = c->S0, nm = c->res.n_merges;
    std::vector<uint32_t> log, hcount;
    if ((rc = fetch(c, c->merges, (size_t)nm * 3, log)) || (rc = fetch(c, c->hcount, S0 + 1, hcount))) return rc;
    if ((rc = eval_truth(c, truth_point_labels))) return rc;
    uint32_t *d_root, *d_incl;
    ENSURE(c->eroot, uint32_t, S0 + 1, d_root); ENSURE(c->eincl, uint32_t, S0 + 1, d_incl);
    std::vector<uint32_t> parent(S0 + 1), root(S0 + 1), incl(S0 + 1);
    for (uint32_t h = 0; h <= S0; ++h) parent[h] = h;
    std::map<float, f3ds_performance> all;
    uint32_t done = 0;
    for (float t : ts) {
        while (done < nm) {
            float w; memcpy(&w, &log[(size_t)done * 3 + 2], 4);
            if (!(w < t)) break;
            parent[log[(size_t)done * 3 + 1]] = log[(size_t)done * 3];
            done++;
        }
        uint32_t K = 0;
        for (uint32_t h = 0; h <= S0; ++h) {
            uint32_t r = h;
            while (parent[r] != r) r = parent[r];
            root[h] = r;
            if (h > 0 && hcount[h] && r == h) K++;
            incl[h] = K;
        }
        HIPCHECK(hipMemcpy(d_root, root.data(), (size_t)(S0 + 1) * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(d_incl, incl.data(), (size_t)(S0 + 1) * 4, hipMemcpyHostToDevice));
        f3ds_performance pf;
        if ((rc = eval_scores(c, d_root, d_incl, K, &pf))) return rc;
        all.insert({t, pf});
    }
    float bt = 0; f3ds_performance bp; memset(&bp, 0, sizeof bp);
    for (auto& kv : all) if (kv.second.fscore > bp.fscore) { bp = kv.second; bt = kv.first; }     // best_thresh, :748-774
    size_t k = 0;
    for (auto& kv : all) { if (k < cap) { if (thresholds) thresholds[k] = kv.first; if (scores) scores[k] = kv.second; } k++; }
    if (n_out) *n_out = k;
    if (best_threshold) *best_threshold = bt;
    if (best_score) *best_score = bp;
    p.threshold = bt;
    return f3ds_recluster(c, &p, point_labels, labels_on_device, result);
}
