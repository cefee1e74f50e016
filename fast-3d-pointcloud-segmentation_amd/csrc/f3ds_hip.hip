// f3ds_hip.hip -- the MI355X (gfx950) device pipeline behind include/f3ds.h.
//
// Stage map (reference citations are in include/f3ds.h, csrc/f3ds_numerics.h, csrc/f3ds_algo.h):
//   0 voxelise   k_bbox -> k_grid -> k_keys -> radix sort -> k_heads/scan/k_segstart -> k_voxel_accum
//   1 neighbours k_neighbors (hash probe of the 27 cells), k_normals (two-ring ordered covariance)
//   2 seeds      k_chunkbox, k_seed_grow, k_seed_keys, radix sort, k_cell_hash, k_seed_nn, k_seed_filter
//   3 sweeps     per sweep: k_ghost_relink, k_sweep_R, k_sweep_claim, k_centroid
//   4 summaries  k_sv_fill (payload rows + ordered leaf sums), k_edges, radix sort, k_edge_init,
//                k_edge_deltas, k_lambda / k_cdf, k_edge_weights
//   5 merge      k_merge (one persistent workgroup), k_roots
//   6 labels     k_region_rank (+scan), k_point_labels
//
// Layout in HBM: points stay as the caller's 16-byte records (one global_load_dwordx4 per lane);
// everything per voxel is SoA rows of 12 floats (48 B, 16-B aligned: xyz rgb normal pad) so a
// lane reads a neighbour with three dwordx4 loads; per-voxel neighbour table is V x 27 int32.
// Float summation order is the reference's everywhere: parallel across outputs, sequential in
// reference order inside each reduction.  Built with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/f3ds.h"
#include "f3ds_algo.h"
#include "f3ds_glasbey.h"

using namespace f3ds;

namespace {

thread_local std::string g_last_hip_error;

#define HIPCHECK(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            g_last_hip_error = std::string(#expr) + ": " + hipGetErrorString(e_);               \
            return F3DS_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
};

struct P16 { float x, y, z; uint32_t rgba; };

// small block of device scalars the host reads back at the few sync points
struct DevCounters {
    unsigned long long n_finite;
    uint32_t bbox[6];        // order-preserving encodings of min xyz, max xyz
    uint32_t bbox_any;
    uint32_t n_valid;        // points with a valid voxel key
    uint32_t n_voxels;
    uint32_t n_cells;
    uint32_t n_seeds;
    uint32_t n_edges;
    uint32_t n_alive;        // non-empty supervoxels
    uint32_t n_merges;
    uint32_t n_regions;
    uint32_t n_events;
    int error;               // first error raised on the device
    int r_overflow;
    float lambda;
    uint32_t seg_count;      // scratch for generic segmenting
    uint32_t n_ghosts;       // helpers that still hold a ghost leaf (sweeps)
    uint32_t pad[2];
};

constexpr uint64_t HASH_EMPTY = 0xFFFFFFFFFFFFFFFFull;

__host__ __device__ inline uint64_t hash64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ inline uint32_t enc_f32(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float dec_f32(uint32_t e) {
    uint32_t b = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    float f; memcpy(&f, &b, 4); return f;
}
__device__ inline int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ inline uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// ------------------------------------------------------------------------------------------------
// generic inclusive scan of uint32 (three launches; any n)
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ inline uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += t;
    }
    return v;
}
// inclusive scan across a block of BLOCK threads (BLOCK multiple of 64, <= 1024); returns the
// inclusive prefix of `v`, *total gets the block sum
template <int BLOCK>
__device__ inline uint32_t block_incl_scan(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wsum[BLOCK / 64];
    __shared__ uint32_t wtot;
    uint32_t inc = wave_incl_scan(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 63) wsum[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < BLOCK / 64; ++i) { uint32_t t = wsum[i]; wsum[i] = run; run += t; }
        wtot = run;
    }
    __syncthreads();
    *total = wtot;
    return inc + wsum[w];
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(const uint32_t* in, uint32_t* out, uint32_t* tile_sums, uint32_t n) {
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = (base + i < n) ? in[base + i] : 0u; s += v[i]; }
    uint32_t tot;
    uint32_t inc = block_incl_scan<SCAN_THREADS>(s, &tot);
    uint32_t run = inc - s;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { run += v[i]; if (base + i < n) out[base + i] = run; }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
// exclusive scan of m values by one block (in place)
__global__ __launch_bounds__(1024) void k_scan_single(uint32_t* data, uint32_t m) {
    const uint32_t per = (m + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per;
    const uint32_t hi = lo + per < m ? lo + per : m;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += data[i];
    uint32_t tot;
    uint32_t inc = block_incl_scan<1024>(s, &tot);
    uint32_t run = inc - s;
    for (uint32_t i = lo; i < hi; ++i) { uint32_t t = data[i]; data[i] = run; run += t; }
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_add(uint32_t* out, const uint32_t* tile_offsets, uint32_t n) {
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    const uint32_t off = tile_offsets[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) out[base + i] += off;
}

// ------------------------------------------------------------------------------------------------
// stable LSD radix sort of (uint64 key, uint32 value) pairs, up to 8 bits per pass.
// One workgroup owns a contiguous tile; inside the tile each wave owns a contiguous quarter and
// walks it in 64-element strips, so "earlier in memory" == "earlier in (wave, strip, lane)" and
// ranks from ballots are stable.
// ------------------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_STRIPS = 16;                         // strips of 64 per wave
constexpr int RS_TILE = RS_THREADS * RS_STRIPS;       // 4096 keys per workgroup

__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const uint64_t* keys, uint32_t n, int shift, int bits, uint32_t* hist,
                                                         uint32_t nblocks) {
    __shared__ uint32_t lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t wbase = blockIdx.x * RS_TILE + (threadIdx.x >> 6) * (64 * RS_STRIPS);
#pragma unroll
    for (int s = 0; s < RS_STRIPS; ++s) {
        uint32_t i = wbase + s * 64 + lane_id();
        if (i < n) atomicAdd(&lh[(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x < (1u << bits)) hist[threadIdx.x * nblocks + blockIdx.x] = lh[threadIdx.x];
}
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(const uint64_t* keys, const uint32_t* vals, uint64_t* keys_out,
                                                            uint32_t* vals_out, uint32_t n, int shift, int bits,
                                                            const uint32_t* hist_scanned, uint32_t nblocks) {
    __shared__ uint32_t wcount[RS_WAVES][256];
    const uint32_t mask = (1u << bits) - 1u;
    const int w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&wcount[0][0])[i] = 0;
    __syncthreads();
    const uint32_t wbase = blockIdx.x * RS_TILE + w * (64 * RS_STRIPS);
    uint64_t k[RS_STRIPS];
#pragma unroll
    for (int s = 0; s < RS_STRIPS; ++s) {
        uint32_t i = wbase + s * 64 + lane_id();
        k[s] = i < n ? keys[i] : 0ull;
        if (i < n) atomicAdd(&wcount[w][(uint32_t)(k[s] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x < (1u << bits)) {   // digit d: global base of this tile, then per-wave starts
        uint32_t run = hist_scanned[threadIdx.x * nblocks + blockIdx.x];
        for (int ww = 0; ww < RS_WAVES; ++ww) { uint32_t t = wcount[ww][threadIdx.x]; wcount[ww][threadIdx.x] = run; run += t; }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < RS_STRIPS; ++s) {
        uint32_t i = wbase + s * 64 + lane_id();
        const bool valid = i < n;
        const uint32_t d = (uint32_t)(k[s] >> shift) & mask;
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
            uint64_t m = __ballot(valid && ((d >> b) & 1u));
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        uint32_t pos = 0;
        if (valid) pos = wcount[w][d] + (uint32_t)__popcll(peers & lanemask_lt());
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lanemask_lt()) == 0ull) wcount[w][d] += (uint32_t)__popcll(peers);   // lowest peer advances the cursor
        __builtin_amdgcn_wave_barrier();
        if (valid) { keys_out[pos] = k[s]; vals_out[pos] = vals[i]; }
    }
}


// ------------------------------------------------------------------------------------------------
// stage 0: voxelise
// ------------------------------------------------------------------------------------------------
struct FrameArgs {
    int use_transform, fold_negative_z, leaf_order;
    float voxel_res, seed_res, w_color, w_spatial, w_normal;
};

__device__ inline P16 load_point(const P16* pts, size_t i) {
    const uint4 q = reinterpret_cast<const uint4*>(pts)[i];     // one global_load_dwordx4
    P16 p;
    p.x = __uint_as_float(q.x); p.y = __uint_as_float(q.y); p.z = __uint_as_float(q.z); p.rgba = q.w;
    return p;
}

// bounding box of the transformed finite points + count of finite input points
__global__ __launch_bounds__(256) void k_bbox(const P16* pts, uint32_t n, FrameArgs fa, DevCounters* dc) {
    float mn[3] = {F3DS_FLT_MAX, F3DS_FLT_MAX, F3DS_FLT_MAX}, mx[3] = {-F3DS_FLT_MAX, -F3DS_FLT_MAX, -F3DS_FLT_MAX};
    uint32_t nfin = 0, any = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        P16 p = load_point(pts, i);
        float x = p.x, y = p.y, z = p.z;
        n_prelude(z, fa.fold_negative_z);
        if (n_finite3(x, y, z)) nfin++;
        n_transform(x, y, z, fa.use_transform);
        if (!n_finite3(x, y, z)) continue;
        any = 1;
        if (x < mn[0]) mn[0] = x;
        if (y < mn[1]) mn[1] = y;
        if (z < mn[2]) mn[2] = z;
        if (x > mx[0]) mx[0] = x;
        if (y > mx[1]) mx[1] = y;
        if (z > mx[2]) mx[2] = z;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        for (int a = 0; a < 3; ++a) {
            float t = __shfl_xor(mn[a], d, 64); if (t < mn[a]) mn[a] = t;
            t = __shfl_xor(mx[a], d, 64); if (t > mx[a]) mx[a] = t;
        }
        nfin += __shfl_xor(nfin, d, 64);
        any |= __shfl_xor(any, d, 64);
    }
    __shared__ float smn[4][3], smx[4][3];
    __shared__ uint32_t sfin[4], sany[4];
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) { for (int a = 0; a < 3; ++a) { smn[w][a] = mn[a]; smx[w][a] = mx[a]; } sfin[w] = nfin; sany[w] = any; }
    __syncthreads();
    if (threadIdx.x == 0) {      // one set of atomics per workgroup: ~13 ns each on one address
        for (int ww = 1; ww < 4; ++ww) {
            for (int a = 0; a < 3; ++a) { if (smn[ww][a] < mn[a]) mn[a] = smn[ww][a]; if (smx[ww][a] > mx[a]) mx[a] = smx[ww][a]; }
            nfin += sfin[ww]; any |= sany[ww];
        }
        if (any) {
            for (int a = 0; a < 3; ++a) { atomicMin(&dc->bbox[a], enc_f32(mn[a])); atomicMax(&dc->bbox[3 + a], enc_f32(mx[a])); }
            atomicOr(&dc->bbox_any, 1u);
        }
        if (nfin) atomicAdd(&dc->n_finite, (unsigned long long)nfin);
    }
}
__global__ void k_grid(DevCounters* dc, float voxel_res, GridInfo* g) {
    if (threadIdx.x || blockIdx.x) return;
    GridInfo t;
    if (!dc->bbox_any) {
        for (int a = 0; a < 3; ++a) { t.min[a] = 0; t.max[a] = 0; }
        t.res = (double)voxel_res; t.depth = 0; t.max_key = 0; t.error = 0; t.empty = 1;
    } else {
        float mn[3], mx[3];
        for (int a = 0; a < 3; ++a) { mn[a] = dec_f32(dc->bbox[a]); mx[a] = dec_f32(dc->bbox[3 + a]); }
        n_grid_from_bbox(mn, mx, voxel_res, t);
        if (t.error) dc->error = t.error;
    }
    *g = t;
}
// Morton key per point (leaf order), invalid points get the one key above every valid one
__global__ __launch_bounds__(256) void k_keys(const P16* pts, uint32_t n, FrameArgs fa, const GridInfo* gp, uint64_t* keys, uint32_t* vals) {
    const GridInfo g = *gp;
    const uint64_t invalid = 1ull << (3 * g.depth);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        P16 p = load_point(pts, i);
        float x = p.x, y = p.y, z = p.z;
        n_prelude(z, fa.fold_negative_z);
        uint64_t key = invalid;
        if (n_finite3(x, y, z) && !g.empty) {
            unsigned k[3];
            n_point_key(g, x, y, z, fa.use_transform, k);
            key = n_morton(k[0], k[1], k[2], g.depth);
            if (fa.leaf_order == 1) key = (~key) & (invalid - 1ull);
        }
        keys[i] = key; vals[i] = i;
    }
}
// segment heads of a sorted key array; keys >= limit are "no segment"
__global__ __launch_bounds__(256) void k_heads(const uint64_t* keys, uint32_t n, uint64_t limit, uint32_t* flags) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t k = keys[i];
        flags[i] = (k < limit && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
    }
}
// seg_start[s] = first sorted position of segment s; seg_start[nseg] = number of valid keys
__global__ __launch_bounds__(256) void k_segstart(const uint64_t* keys, const uint32_t* flags, const uint32_t* incl, uint32_t n, uint64_t limit,
                                                 uint32_t* seg_start, uint32_t* nseg_out, uint32_t* nvalid_out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (flags[i]) seg_start[incl[i] - 1u] = i;
        if (keys[i] < limit && (i + 1 == n || keys[i + 1] >= limit)) { seg_start[incl[i]] = i + 1u; *nseg_out = incl[i]; *nvalid_out = i + 1u; }
    }
}
// one lane per voxel: ordered sums over its points (input order), centroid, key, hash insert
__global__ __launch_bounds__(256) void k_voxel_accum(const P16* pts, const uint64_t* keys, const uint32_t* vals, const uint32_t* seg_start,
                                                    const DevCounters* dc, FrameArgs fa, const GridInfo* gp, uint32_t* vkey, uint32_t* vcount,
                                                    float* vf, int* pt_voxel, uint64_t* hkeys, uint32_t* hvals, uint32_t hmask) {
    const uint32_t V = dc->n_voxels;
    const int depth = gp->depth;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x) {
        const uint32_t s = seg_start[v], e = seg_start[v + 1];
        float sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0;
        for (uint32_t i = s; i < e; ++i) {
            const uint32_t idx = vals[i];
            P16 p = load_point(pts, idx);
            float z = p.z; n_prelude(z, fa.fold_negative_z);
            sx += p.x; sy += p.y; sz += z;
            sr += (float)((p.rgba >> 16) & 255u); sg += (float)((p.rgba >> 8) & 255u); sb += (float)(p.rgba & 255u);
            pt_voxel[idx] = (int)v;
        }
        const uint32_t cnt = e - s;
        const float c = (float)cnt;
        float4* row = reinterpret_cast<float4*>(vf + (size_t)v * 12);
        row[0] = make_float4(sx / c, sy / c, sz / c, sr / c);
        row[1] = make_float4(sg / c, sb / c, 0.0f, 0.0f);
        row[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        vcount[v] = cnt;
        uint64_t code = keys[s];
        if (fa.leaf_order == 1) code = (~code) & ((1ull << (3 * depth)) - 1ull);
        unsigned k[3];
        n_demorton(code, depth, k);
        vkey[v * 3] = k[0]; vkey[v * 3 + 1] = k[1]; vkey[v * 3 + 2] = k[2];
        const uint64_t pk = n_pack_key(k[0], k[1], k[2]);
        uint32_t h = (uint32_t)hash64(pk) & hmask;
        for (;;) {
            unsigned long long old = atomicCAS((unsigned long long*)&hkeys[h], (unsigned long long)HASH_EMPTY, (unsigned long long)pk);
            if (old == HASH_EMPTY || old == pk) { hvals[h] = v; break; }
            h = (h + 1u) & hmask;
        }
    }
}
__device__ inline int hash_find(const uint64_t* hkeys, const uint32_t* hvals, uint32_t hmask, uint64_t pk) {
    uint32_t h = (uint32_t)hash64(pk) & hmask;
    for (;;) {
        uint64_t k = hkeys[h];
        if (k == pk) return (int)hvals[h];
        if (k == HASH_EMPTY) return -1;
        h = (h + 1u) & hmask;
    }
}

// ------------------------------------------------------------------------------------------------
// stage 1: neighbour table and voxel normals
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_neighbors(const uint32_t* vkey, const DevCounters* dc, const GridInfo* gp, const uint64_t* hkeys,
                                                  const uint32_t* hvals, uint32_t hmask, int* nbr, int* nbrT) {
    const uint32_t V = dc->n_voxels;
    const uint32_t total = V * 27u;
    const unsigned max_key = gp->max_key;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const uint32_t v = t / 27u, s = t - v * 27u;
        const int d[3] = {(int)(s / 9u) - 1, (int)((s / 3u) % 3u) - 1, (int)(s % 3u) - 1};
        bool ok = true; unsigned k[3];
        for (int a = 0; a < 3; ++a) {
            long long q = (long long)vkey[v * 3 + a] + d[a];
            if (q < 0 || q > (long long)max_key) ok = false;
            k[a] = (unsigned)q;
        }
        const int u = ok ? hash_find(hkeys, hvals, hmask, n_pack_key(k[0], k[1], k[2])) : -1;
        nbr[t] = u;                              // row-major: a voxel's 27 slots together (normals, adjacency)
        nbrT[(size_t)s * V + v] = u;             // slot-major: coalesced across consecutive voxels (sweeps)
    }
}
__device__ inline void cov_add(float acc[9], const float4 q) {
    acc[0] += q.x * q.x; acc[1] += q.x * q.y; acc[2] += q.x * q.z;
    acc[3] += q.y * q.y; acc[4] += q.y * q.z; acc[5] += q.z * q.z;
    acc[6] += q.x; acc[7] += q.y; acc[8] += q.z;
}
__global__ __launch_bounds__(256) void k_normals(float* vf, const int* nbr, const DevCounters* dc) {
    const uint32_t V = dc->n_voxels;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x) {
        float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned cnt = 1;
        const float4 self = *reinterpret_cast<const float4*>(vf + (size_t)v * 12);
        cov_add(acc, self);
        for (int s = 0; s < 27; ++s) {
            const int u = nbr[(size_t)v * 27 + s];
            if (u < 0) continue;
            cov_add(acc, *reinterpret_cast<const float4*>(vf + (size_t)u * 12)); cnt++;
            for (int s2 = 0; s2 < 27; ++s2) {
                const int u2 = nbr[(size_t)u * 27 + s2];
                if (u2 >= 0) { cov_add(acc, *reinterpret_cast<const float4*>(vf + (size_t)u2 * 12)); cnt++; }
            }
        }
        const float vp[3] = {self.x, self.y, self.z};
        float n4[4];
        n_plane_normal(acc, cnt, vp, n4);
        // normals live in a separate array until every voxel has read the centroids
        float* nout = vf + (size_t)v * 12 + 6;
        nout[0] = n4[0]; nout[1] = n4[1]; nout[2] = n4[2];
    }
}

// ------------------------------------------------------------------------------------------------
// stage 2: seeds
// ------------------------------------------------------------------------------------------------
constexpr int SEED_CHUNK = 256;
__global__ __launch_bounds__(SEED_CHUNK) void k_chunkbox(const float* vf, const DevCounters* dc, float* boxes) {
    const uint32_t V = dc->n_voxels;
    const uint32_t v = blockIdx.x * SEED_CHUNK + threadIdx.x;
    if (blockIdx.x * SEED_CHUNK >= V) return;
    float mn[3] = {F3DS_FLT_MAX, F3DS_FLT_MAX, F3DS_FLT_MAX}, mx[3] = {-F3DS_FLT_MAX, -F3DS_FLT_MAX, -F3DS_FLT_MAX};
    if (v < V) { const float* p = vf + (size_t)v * 12; for (int a = 0; a < 3; ++a) { mn[a] = p[a]; mx[a] = p[a]; } }
    __shared__ float smn[SEED_CHUNK / 64][3], smx[SEED_CHUNK / 64][3];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        for (int a = 0; a < 3; ++a) {
            float t = __shfl_xor(mn[a], d, 64); if (t < mn[a]) mn[a] = t;
            t = __shfl_xor(mx[a], d, 64); if (t > mx[a]) mx[a] = t;
        }
    if (lane_id() == 0) for (int a = 0; a < 3; ++a) { smn[threadIdx.x >> 6][a] = mn[a]; smx[threadIdx.x >> 6][a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SEED_CHUNK / 64; ++w)
            for (int a = 0; a < 3; ++a) { if (smn[w][a] < mn[a]) mn[a] = smn[w][a]; if (smx[w][a] > mx[a]) mx[a] = smx[w][a]; }
        for (int a = 0; a < 3; ++a) { boxes[blockIdx.x * 6 + a] = mn[a]; boxes[blockIdx.x * 6 + 3 + a] = mx[a]; }
    }
}
// replay of OctreePointCloud::adoptBoundingBoxToPoint over the voxel centroids in leaf order:
// the cube only changes when a point falls outside it, so look for the first such point (chunk
// boxes first, then the points of that chunk), grow, and continue behind it.
__global__ __launch_bounds__(1024) void k_seed_grow(const float* vf, const float* boxes, DevCounters* dc, float seed_res, SeedGrid* out) {
    __shared__ SeedGrid g;
    __shared__ uint32_t cursor;
    __shared__ uint32_t found;
    const uint32_t V = dc->n_voxels;
    const uint32_t C = (V + SEED_CHUNK - 1) / SEED_CHUNK;
    if (threadIdx.x == 0) { a_seed_init(g, seed_res); cursor = 0; }
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) found = 0xFFFFFFFFu;
        __syncthreads();
        const uint32_t c0 = cursor / SEED_CHUNK;
        if (cursor >= V || g.error) break;
        uint32_t mine = 0xFFFFFFFFu;
        for (uint32_t c = c0 + threadIdx.x; c < C; c += 1024u)
            if (a_seed_box_violates(g, boxes + c * 6, boxes + c * 6 + 3)) { mine = c; break; }
        if (mine != 0xFFFFFFFFu) atomicMin(&found, mine);
        __syncthreads();
        const uint32_t cstar = found;
        __syncthreads();
        if (cstar == 0xFFFFFFFFu) break;
        if (threadIdx.x == 0) found = 0xFFFFFFFFu;
        __syncthreads();
        {
            const uint32_t v = cstar * SEED_CHUNK + threadIdx.x;
            if (threadIdx.x < SEED_CHUNK && v < V && v >= cursor && a_seed_violates(g, vf + (size_t)v * 12)) atomicMin(&found, v);
        }
        __syncthreads();
        const uint32_t istar = found;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (istar == 0xFFFFFFFFu) cursor = (cstar + 1u) * SEED_CHUNK;
            else { a_seed_grow(g, (int)istar, vf + (size_t)istar * 12); cursor = istar + 1u; }
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) { *out = g; if (g.error) dc->error = g.error; }
}
__global__ __launch_bounds__(256) void k_seed_keys(const float* vf, const DevCounters* dc, const SeedGrid* gp, uint32_t* ckey, uint64_t* keys, uint32_t* vals) {
    __shared__ SeedGrid g;
    if (threadIdx.x == 0) g = *gp;
    __syncthreads();
    const uint32_t V = dc->n_voxels;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x) {
        unsigned k[3];
        a_seed_key(g, (int)v, vf + (size_t)v * 12, k);
        ckey[v * 3] = k[0]; ckey[v * 3 + 1] = k[1]; ckey[v * 3 + 2] = k[2];
        keys[v] = n_morton(k[0], k[1], k[2], g.depth);
        vals[v] = v;
    }
}
__global__ __launch_bounds__(256) void k_cell_hash(const uint32_t* ckey, const uint32_t* sorted_vox, const uint32_t* cell_start, const DevCounters* dc,
                                                  uint64_t* hkeys, uint32_t* hvals, uint32_t hmask) {
    const uint32_t C = dc->n_cells;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        const uint32_t v = sorted_vox[cell_start[c]];
        const uint64_t pk = n_pack_key(ckey[v * 3], ckey[v * 3 + 1], ckey[v * 3 + 2]);
        uint32_t h = (uint32_t)hash64(pk) & hmask;
        for (;;) {
            unsigned long long old = atomicCAS((unsigned long long*)&hkeys[h], (unsigned long long)HASH_EMPTY, (unsigned long long)pk);
            if (old == HASH_EMPTY || old == pk) { hvals[h] = c; break; }
            h = (h + 1u) & hmask;
        }
    }
}
// one wave per occupied seed cell: exact nearest voxel to the cell centre (3x3x3 cell block)
__global__ __launch_bounds__(64) void k_seed_nn(const float* vf, const uint32_t* ckey, const uint32_t* sorted_vox, const uint32_t* cell_start,
                                               const DevCounters* dc, const SeedGrid* gp, const uint64_t* hkeys, const uint32_t* hvals, uint32_t hmask,
                                               int* seed_orig) {
    const uint32_t c = blockIdx.x;
    if (c >= dc->n_cells) return;
    const uint32_t v0 = sorted_vox[cell_start[c]];
    const unsigned key[3] = {ckey[v0 * 3], ckey[v0 * 3 + 1], ckey[v0 * 3 + 2]};
    float centre[3];
    for (int a = 0; a < 3; ++a) centre[a] = (float)(((double)key[a] + 0.5f) * gp->res + gp->min[a]);
    float bd = F3DS_FLT_MAX; int best = 0x7fffffff;
    for (int s = 0; s < 27; ++s) {
        long long x = (long long)key[0] + s / 9 - 1, y = (long long)key[1] + (s / 3) % 3 - 1, z = (long long)key[2] + s % 3 - 1;
        if (x < 0 || y < 0 || z < 0) continue;
        const int cc = hash_find(hkeys, hvals, hmask, n_pack_key((unsigned)x, (unsigned)y, (unsigned)z));
        if (cc < 0) continue;
        for (uint32_t i = cell_start[cc] + lane_id(); i < cell_start[cc + 1]; i += 64u) {
            const int j = (int)sorted_vox[i];
            const float d = a_sqdist(centre, vf + (size_t)j * 12);
            if (best == 0x7fffffff || d < bd || (d == bd && j < best)) { bd = d; best = j; }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float od = __shfl_xor(bd, d, 64); const int ob = __shfl_xor(best, d, 64);
        if (ob != 0x7fffffff && (best == 0x7fffffff || od < bd || (od == bd && ob < best))) { bd = od; best = ob; }
    }
    if (lane_id() == 0) seed_orig[c] = best;
}
// one wave per candidate seed: voxels closer than seed_res/2 to the seed voxel
__global__ __launch_bounds__(64) void k_seed_filter(const float* vf, const uint32_t* ckey, const uint32_t* sorted_vox, const uint32_t* cell_start,
                                                   const DevCounters* dc, const uint64_t* hkeys, const uint32_t* hvals, uint32_t hmask,
                                                   const int* seed_orig, float r2, float min_points, uint32_t* keep) {
    const uint32_t c = blockIdx.x;
    if (c >= dc->n_cells) return;
    const int s0 = seed_orig[c];
    const unsigned key[3] = {ckey[s0 * 3], ckey[s0 * 3 + 1], ckey[s0 * 3 + 2]};
    uint32_t num = 0;
    for (int s = 0; s < 27; ++s) {
        long long x = (long long)key[0] + s / 9 - 1, y = (long long)key[1] + (s / 3) % 3 - 1, z = (long long)key[2] + s % 3 - 1;
        if (x < 0 || y < 0 || z < 0) continue;
        const int cc = hash_find(hkeys, hvals, hmask, n_pack_key((unsigned)x, (unsigned)y, (unsigned)z));
        if (cc < 0) continue;
        for (uint32_t i = cell_start[cc] + lane_id(); i < cell_start[cc + 1]; i += 64u)
            if (a_sqdist(vf + (size_t)s0 * 12, vf + (size_t)sorted_vox[i] * 12) < r2) num++;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) num += __shfl_xor(num, d, 64);
    if (lane_id() == 0) keep[c] = ((float)num > min_points) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_seed_compact(const int* seed_orig, const uint32_t* keep, const uint32_t* incl, DevCounters* dc, int* seed_kept) {
    const uint32_t C = dc->n_cells;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        if (keep[c]) seed_kept[incl[c] - 1u] = seed_orig[c];
        if (c + 1 == C) dc->n_seeds = incl[c];
    }
}

// ------------------------------------------------------------------------------------------------
// stage 3: helpers and label-propagation sweeps
// ------------------------------------------------------------------------------------------------
// createSupervoxelHelpers: the LAST helper seeded on a voxel owns it (addLeaf overwrites owner_)
__global__ __launch_bounds__(256) void k_helper_own(const int* seed_kept, uint32_t S0, uint32_t* owner) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < S0; i += gridDim.x * blockDim.x) atomicMax(&owner[seed_kept[i]], i + 1u);
}
__global__ __launch_bounds__(256) void k_helper_init(const int* seed_kept, uint32_t S0, const uint32_t* owner, int* ghost_vox, unsigned char* ghost_active,
                                                    unsigned char* ghost_done, uint32_t* hlo, uint32_t* hhi, uint32_t* hcount, float* hc) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= S0; i += gridDim.x * blockDim.x) {
        for (int k = 0; k < 12; ++k) hc[(size_t)i * 12 + k] = 0.0f;
        ghost_done[i] = 0;
        if (i == 0) { ghost_vox[0] = -1; ghost_active[0] = 0; hlo[0] = 0; hhi[0] = 0; hcount[0] = 0; continue; }
        const int v = seed_kept[i - 1];
        const bool ghost = owner[v] != i;
        ghost_vox[i] = ghost ? v : -1;
        ghost_active[i] = ghost ? 1 : 0;
        hlo[i] = (uint32_t)v; hhi[i] = (uint32_t)v; hcount[i] = 1;
    }
}
// The sweep kernels are batched: blockIdx.y selects the frame, so the sweeps of a whole batch of
// frames are four dispatches per sweep instead of four per frame and sweep.
struct SweepFrame {
    SweepView sv;                   // sv.owner / sv.dist: the sweep-start state
    unsigned char* R; uint32_t* ownR;
    uint32_t* owner_out; float* dist_out;
    unsigned char *ghost_done, *ghost_active; int* ghost_vox; uint32_t *ghost_head, *ghost_next;
    uint32_t *hlo, *hhi, *hcount; float* hc; DevCounters* dc; uint32_t S0;
};
__global__ __launch_bounds__(256) void k_ghost_relink(const SweepFrame* F) {
    const SweepFrame& f = F[blockIdx.y];
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    for (uint32_t h = 1 + threadIdx.x; h <= f.S0; h += blockDim.x) if (f.ghost_vox[h] >= 0) f.ghost_head[f.ghost_vox[h]] = 0u;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t h = 1 + threadIdx.x; h <= f.S0; h += blockDim.x)
        if (f.ghost_active[h]) { f.ghost_next[h] = atomicExch(&f.ghost_head[f.ghost_vox[h]], h); mine++; }
    if (mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0) f.dc->n_ghosts = s_n;
}
__global__ __launch_bounds__(256) void k_sweep_R(const SweepFrame* F, unsigned char tag) {
    const SweepFrame& f = F[blockIdx.y];
    const SweepView s = f.sv;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < s.V; v += gridDim.x * blockDim.x) {
        int overflow = 0;
        const uint32_t o = s.owner[v];
        const bool r = o ? a_eval_R(s, v, f.R, tag, &overflow) : false;
        f.ownR[v] = o | (r ? F3DS_OWNR_RTRUE : 0u);
        if (overflow) f.dc->r_overflow = 1;
    }
}
__global__ __launch_bounds__(256) void k_sweep_claim(const SweepFrame* F) {
    const SweepFrame& f = F[blockIdx.y];
    const SweepView s = f.sv;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < s.V; v += gridDim.x * blockDim.x) {
        uint32_t o; float d;
        a_claim(s, f.ownR, v, &o, &d, f.ghost_done);
        f.owner_out[v] = o; f.dist_out[v] = d;
        if (o != s.owner[v] && o != 0u) { atomicMin(&f.hlo[o], (uint32_t)v); atomicMax(&f.hhi[o], (uint32_t)v); }
    }
}
// one wave per helper: SupervoxelHelper::updateCentroid.  The helper's leaves are the voxels it
// owns inside its ordinal window [lo,hi] (plus its ghost leaf); they are visited in ascending
// ordinal, i.e. std::set<leaf, compareLeaves> order, 64 candidates at a time.
__global__ __launch_bounds__(64) void k_centroid(const SweepFrame* F) {
    __shared__ __attribute__((aligned(16))) float tile[64][12];
    const SweepFrame& f = F[blockIdx.y];
    const uint32_t h = blockIdx.x + 1u;
    if (h > f.S0) return;
    const float* vf = f.sv.vf; const uint32_t* owner = f.owner_out;
    const int lane = lane_id();
    bool gact = f.ghost_active[h] && !f.ghost_done[h];
    const int gv = gact ? f.ghost_vox[h] : -1;
    const uint32_t lo = f.hlo[h], hi = f.hhi[h];
    float acc = 0.0f;
    uint32_t count = 0;
    for (uint32_t base = lo & ~63u; base <= hi; base += 64u) {
        const uint32_t v = base + lane;
        const bool m = v >= lo && v <= hi && (owner[v] == h || (int)v == gv);
        const uint64_t mask = __ballot(m);
        if (!mask) continue;
        const int rank = __popcll(mask & lanemask_lt());
        const int cnt = __popcll(mask);
        if (m) {
            const float4* row = reinterpret_cast<const float4*>(vf + (size_t)v * 12);
            float4* t = reinterpret_cast<float4*>(&tile[rank][0]);
            t[0] = row[0]; t[1] = row[1]; t[2] = row[2];
        }
        __syncthreads();
        if (lane < 9) for (int j = 0; j < cnt; ++j) acc += tile[j][lane];
        count += (uint32_t)cnt;
        __syncthreads();
    }
    if (lane == 0) { f.ghost_active[h] = gact ? 1 : 0; f.ghost_done[h] = 0; f.hcount[h] = count; }
    if (count == 0) return;
    float sum[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) sum[k] = __shfl(acc, k, 64);
    if (lane == 0) {
        float row[12];
        a_centroid_finish(sum, count, row);
        for (int k = 0; k < 12; ++k) f.hc[(size_t)h * 12 + k] = row[k];
    }
}
// ------------------------------------------------------------------------------------------------
// stage 4: supervoxel payload, adjacency, initial weights
// ------------------------------------------------------------------------------------------------
// one wave per helper (makeSupervoxels): payload rows of its leaves in leaf order, the ordered
// sums of the leaf (what computeCentroid / computePointNormal / mean_color would accumulate over
// voxels_), and the initial region record.
__global__ __launch_bounds__(64) void k_sv_fill(const float* vf, const uint32_t* owner, uint32_t S0, const uint32_t* hlo, const uint32_t* hhi,
                                               const int* ghost_vox, const unsigned char* ghost_active, const uint32_t* hcount, const uint32_t* loff,
                                               const float* hc, float* rows, int* row_voxel, float* racc0, uint32_t* rcnt0, float* rrec0,
                                               unsigned char* ralive0, DevCounters* dc) {
    __shared__ __attribute__((aligned(16))) float tile[64][12];
    __shared__ int tile_v[64];
    const uint32_t h = blockIdx.x + 1u;
    if (h > S0) return;
    const int lane = lane_id();
    const uint32_t len = hcount[h];
    if (len == 0) {
        if (lane == 0) { ralive0[h] = 0; rcnt0[h] = 0; }
        if (lane < 12) racc0[(size_t)h * 12 + lane] = 0.0f;
        if (lane < 16) rrec0[(size_t)h * 16 + lane] = 0.0f;
        return;
    }
    const uint32_t off = loff[h];
    const int gv = ghost_active[h] ? ghost_vox[h] : -1;
    const uint32_t lo = hlo[h], hi = hhi[h];
    float acc = 0.0f;
    uint32_t done = 0;
    const int ia = lane < 6 ? (lane < 3 ? 0 : (lane < 5 ? 1 : 2)) : (lane < 9 ? lane - 6 : lane - 6);
    const int ib = lane < 6 ? (lane < 3 ? lane : (lane < 5 ? lane - 2 : 2)) : 0;
    for (uint32_t base = lo & ~63u; base <= hi; base += 64u) {
        const uint32_t v = base + lane;
        const bool m = v >= lo && v <= hi && (owner[v] == h || (int)v == gv);
        const uint64_t mask = __ballot(m);
        if (!mask) continue;
        const int rank = __popcll(mask & lanemask_lt());
        const int cnt = __popcll(mask);
        if (m) {
            const float4* row = reinterpret_cast<const float4*>(vf + (size_t)v * 12);
            float4* t = reinterpret_cast<float4*>(&tile[rank][0]);
            t[0] = row[0]; t[1] = row[1]; t[2] = row[2];
            tile_v[rank] = (int)v;
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            if (lane < 12) {
                float val;
                if (lane < 6) val = tile[j][ia] * tile[j][ib];
                else if (lane < 9) val = tile[j][ia];
                else val = (float)((uint32_t)tile[j][ia] & 255u);
                rows[(size_t)(off + done + j) * 12 + lane] = val;
                if (lane < 9) acc += val;
                else { const float count = (float)(done + j + 1u); const float inv = 1 / count; acc = acc + inv * (val - acc); }
            }
            if (lane == 0) row_voxel[off + done + j] = tile_v[j];
        }
        done += (uint32_t)cnt;
        __syncthreads();
    }
    if (lane < 12) racc0[(size_t)h * 12 + lane] = acc;
    const float mr = __shfl(acc, 9, 64), mg = __shfl(acc, 10, 64), mb = __shfl(acc, 11, 64);
    if (lane == 0) {
        rcnt0[h] = len; ralive0[h] = 1;
        atomicAdd(&dc->n_alive, 1u);
        float rec[16];
        const float* c = hc + (size_t)h * 12;
        rec[0] = c[0]; rec[1] = c[1]; rec[2] = c[2]; rec[3] = c[6]; rec[4] = c[7]; rec[5] = c[8];
        rec[6] = mr; rec[7] = mg; rec[8] = mb;
        n_rgb2lab(rec + 6, rec + 9);
        rec[12] = rec[13] = rec[14] = rec[15] = 0.0f;
        for (int k = 0; k < 16; ++k) rrec0[(size_t)h * 16 + k] = rec[k];
    }
}
__device__ inline void edge_insert(uint64_t key, uint64_t* hkeys, uint32_t hmask, uint64_t* ekeys, uint32_t ecap, DevCounters* dc) {
    uint32_t hh = (uint32_t)hash64(key) & hmask;
    for (uint32_t probes = 0; probes <= hmask; ++probes) {
        unsigned long long old = atomicCAS((unsigned long long*)&hkeys[hh], (unsigned long long)HASH_EMPTY, (unsigned long long)key);
        if (old == key) return;
        if (old == HASH_EMPTY) {
            uint32_t pos = atomicAdd(&dc->n_edges, 1u);
            if (pos < ecap) ekeys[pos] = key; else dc->error = F3DS_ERR_UNSUPPORTED;
            return;
        }
        hh = (hh + 1u) & hmask;
    }
    dc->error = F3DS_ERR_UNSUPPORTED;
}
// getSupervoxelAdjacency + clear_adjacency: pairs (h, o), h < o, seen from a leaf of h
__device__ inline void leaf_edges(uint32_t h, uint32_t v, uint32_t S0, const int* nbr, const uint32_t* owner, uint64_t* hkeys, uint32_t hmask,
                                  uint64_t* ekeys, uint32_t ecap, DevCounters* dc) {
    uint32_t last = 0;
    for (int s = 0; s < 27; ++s) {
        const int u = nbr[(size_t)v * 27 + s];
        if (u < 0) continue;
        const uint32_t o = owner[u];
        if (o && o != h && h < o && o != last) { last = o; edge_insert((uint64_t)h * (S0 + 1ull) + o, hkeys, hmask, ekeys, ecap, dc); }
    }
}
__global__ __launch_bounds__(256) void k_edges(uint32_t V, uint32_t S0, const int* nbr, const uint32_t* owner, uint64_t* hkeys, uint32_t hmask,
                                              uint64_t* ekeys, uint32_t ecap, DevCounters* dc) {
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x) {
        const uint32_t h = owner[v];
        if (h) leaf_edges(h, v, S0, nbr, owner, hkeys, hmask, ekeys, ecap, dc);
    }
}
__global__ __launch_bounds__(256) void k_edges_ghost(uint32_t S0, const int* ghost_vox, const unsigned char* ghost_active, const int* nbr,
                                                    const uint32_t* owner, uint64_t* hkeys, uint32_t hmask, uint64_t* ekeys, uint32_t ecap, DevCounters* dc) {
    for (uint32_t h = 1 + blockIdx.x * blockDim.x + threadIdx.x; h <= S0; h += gridDim.x * blockDim.x)
        if (ghost_active[h]) leaf_edges(h, (uint32_t)ghost_vox[h], S0, nbr, owner, hkeys, hmask, ekeys, ecap, dc);
}
__global__ __launch_bounds__(256) void k_iota(uint32_t* v, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = i;
}
__global__ __launch_bounds__(256) void k_fill_f32(float* v, uint32_t n, float x) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = x;
}
__global__ __launch_bounds__(256) void k_edge_init(const uint64_t* ekeys, uint32_t E, uint32_t S0, uint32_t* ea0, uint32_t* eb0) {
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        ea0[e] = (uint32_t)(ekeys[e] / (S0 + 1ull)); eb0[e] = (uint32_t)(ekeys[e] % (S0 + 1ull));
    }
}
__global__ __launch_bounds__(256) void k_edge_deltas(uint32_t E, const uint32_t* ea, const uint32_t* eb, const float* rrec, int color_metric, int geom_metric,
                                                    float* deltas, uint64_t* skeys, uint32_t* svals) {
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        float dc_, dg_;
        n_delta_c_g(rrec + (size_t)ea[e] * 16, rrec + (size_t)eb[e] * 16, color_metric, geom_metric, &dc_, &dg_);
        deltas[e * 2] = dc_; deltas[e * 2 + 1] = dg_;
        if (skeys) {   // one sort orders both multisets: (which << 32 | key)
            skeys[e] = (uint64_t)n_weight_key(dc_); svals[e] = e * 2u;
            skeys[E + e] = (1ull << 32) | (uint64_t)n_weight_key(dg_); svals[E + e] = e * 2u + 1u;
        }
    }
}
// Clustering::deltas_mean over the ascending multisets, then lambda (src/clustering.cpp:267-273).
// The running mean is a serial chain; the 64 lanes fetch the next 64 sorted values of both
// multisets together and lanes 0 (colour) / 1 (geometry) consume them through readlane.
__global__ __launch_bounds__(64) void k_lambda(uint32_t E, const float* deltas, const uint32_t* svals, DevCounters* dc) {
    const int lane = lane_id();
    float count = 0, mean_d = 0;
    for (uint32_t base = 0; base < E; base += 64u) {
        const uint32_t i = base + lane;
        const float vc = i < E ? deltas[svals[i]] : 0.0f;
        const float vg = i < E ? deltas[svals[(size_t)E + i]] : 0.0f;
        const int nb = E - base < 64u ? (int)(E - base) : 64;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            if (j < nb) {
                const float dcj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vc), j));
                const float dgj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vg), j));
                const float d = lane == 0 ? dcj : dgj;
                count++;
                mean_d = mean_d + (1 / count) * (d - mean_d);
            }
        }
    }
    const float mean_c = __shfl(mean_d, 0, 64), mean_g = __shfl(mean_d, 1, 64);
    if (lane == 0) dc->lambda = mean_g / (mean_c + mean_g);
}
// Clustering::compute_cdf (src/clustering.cpp:289-314) for delta_c (w=0) and delta_g (w=1)
__global__ __launch_bounds__(256) void k_cdf_hist(uint32_t E, const float* deltas, int bins, uint32_t* hist, DevCounters* dc) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < 2u * E; t += gridDim.x * blockDim.x) {
        const uint32_t w = t / E, e = t - w * E;
        const float d = deltas[e * 2 + w];
        short bin = (short)__builtin_floorf(d * (float)(short)bins);
        if (bin == (short)bins) bin--;
        if (bin < 0 || bin >= (short)bins) { dc->error = F3DS_ERR_EQ_BIN; continue; }
        atomicAdd(&hist[w * (uint32_t)bins + (uint32_t)bin], 1u);
    }
}
__global__ void k_cdf_scan(uint32_t E, int bins, const uint32_t* hist, float* cdf) {
    const int w = threadIdx.x;
    if (w >= 2) return;
    float v = 0;
    for (int i = 0; i < bins; ++i) { v += (float)hist[w * bins + i]; cdf[w * bins + i] = v / (float)(int)E; }
}
struct MergeDev {
    uint32_t E, S0;
    uint32_t *ea, *eb; float* ew; uint32_t* eku; int* ehist; unsigned char* ealive;
    uint32_t *ev_epoch, *ev_key; int* ev_prev; uint32_t ev_cap;
    float *racc, *rrec; uint32_t* rcnt; unsigned char* ralive;
    uint32_t *rhead, *rtail, *lnext, *parent;
    const uint32_t *loff, *llen; const float* rows;
    uint32_t *markA, *markB, *tl;
    uint32_t* merges;
    float threshold;
    MergeParams mp;
    DevCounters* dc;
};
__global__ __launch_bounds__(256) void k_edge_weights(MergeDev m, const float* deltas) {
    MergeParams mp = m.mp;
    if (mp.merging == 1) mp.lambda = m.dc->lambda;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < m.E; e += gridDim.x * blockDim.x) {
        int err = 0;
        const float w = a_tc(mp, deltas[e * 2], &err) + a_tg(mp, deltas[e * 2 + 1], &err);
        if (err) m.dc->error = err;
        m.ew[e] = w; m.eku[e] = n_weight_key(w); m.ehist[e] = (int)e; m.ealive[e] = 1;
        m.ev_epoch[e] = 0u; m.ev_key[e] = m.eku[e]; m.ev_prev[e] = -1;
    }
}
__global__ __launch_bounds__(256) void k_region_reset(uint32_t S0, const uint32_t* hcount, uint32_t* rhead, uint32_t* rtail, uint32_t* lnext, uint32_t* parent,
                                                     uint32_t* markA, uint32_t* markB, uint32_t* pool, uint32_t* rstart, uint32_t* rnleaf, uint32_t* rcap) {
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h <= S0; h += gridDim.x * blockDim.x) {
        const bool alive = h > 0 && hcount[h] > 0;
        rhead[h] = alive ? h : 0u; rtail[h] = alive ? h : 0u; lnext[h] = 0u; parent[h] = h; markA[h] = 0u; markB[h] = 0u;
        pool[h] = h; rstart[h] = h; rnleaf[h] = alive ? 1u : 0u; rcap[h] = 1u;
    }
}

// ------------------------------------------------------------------------------------------------
// stage 5: the merge loop (Clustering::cluster / merge), one persistent workgroup.
// Per merge: block-wide argmin in weight_map order, then wave 0 folds region b's voxels into
// region a's ordered sums and rebuilds a's record while the other waves collect the incident
// edges; duplicates (x adjacent to both a and b) keep the earlier map entry; kept edges get a
// new weight and a history event.
// ------------------------------------------------------------------------------------------------
constexpr int MG_THREADS = 1024;
__device__ inline bool edge_before(const MergeDev& m, uint32_t e, uint32_t f) {
    EdgeHist H{m.ev_epoch, m.ev_key, m.ev_prev};
    return a_edge_before(H, e, m.eku[e], m.ehist[e], f, m.eku[f], m.ehist[f]);
}
__global__ __launch_bounds__(MG_THREADS) void k_merge(MergeDev m) {
    __shared__ int s_wbest[MG_THREADS / 64];
    __shared__ int s_best;
    __shared__ uint32_t s_a, s_b, s_ntouched, s_nmerges, s_nevents;
    __shared__ int s_stop;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    MergeParams mp = m.mp;
    if (mp.merging == 1) mp.lambda = m.dc->lambda;
    if (tid == 0) { s_ntouched = 0; s_nmerges = 0; s_nevents = m.E; s_stop = 0; }
    __syncthreads();
    for (uint32_t epoch = 1;; ++epoch) {
        // ---- next = *weight_map.begin()
        int best = -1;
        for (uint32_t e = tid; e < m.E; e += MG_THREADS)
            if (m.ealive[e] && (best < 0 || edge_before(m, e, (uint32_t)best))) best = (int)e;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int ob = __shfl_xor(best, d, 64);
            if (ob >= 0 && (best < 0 || edge_before(m, (uint32_t)ob, (uint32_t)best))) best = ob;
        }
        if (lane == 0) s_wbest[wave] = best;
        __syncthreads();
        if (tid == 0) {
            int b = -1;
            for (int w = 0; w < MG_THREADS / 64; ++w) { const int ob = s_wbest[w]; if (ob >= 0 && (b < 0 || edge_before(m, (uint32_t)ob, (uint32_t)b))) b = ob; }
            s_best = b;
            if (b < 0 || !(m.ew[b] < m.threshold)) s_stop = 1;
            else {
                s_a = m.ea[b]; s_b = m.eb[b];
                const uint32_t k = s_nmerges++;
                m.merges[k * 3] = s_a; m.merges[k * 3 + 1] = s_b; m.merges[k * 3 + 2] = __float_as_uint(m.ew[b]);
                m.ealive[b] = 0;
            }
        }
        __syncthreads();
        if (s_stop) break;
        const uint32_t a = s_a, b = s_b;
        if (wave == 0) {
            // ---- voxels_new = voxels_a ++ voxels_b: continue a's ordered sums over b's rows
            float acc = lane < 12 ? m.racc[(size_t)a * 12 + lane] : 0.0f;
            uint32_t cnt = m.rcnt[a];
            for (uint32_t leaf = m.rhead[b]; leaf; leaf = m.lnext[leaf]) {
                const uint32_t off = m.loff[leaf], len = m.llen[leaf];
                for (uint32_t j0 = 0; j0 < len; j0 += 16u) {
                    float val[16];
#pragma unroll
                    for (int t = 0; t < 16; ++t) val[t] = (lane < 12 && j0 + t < len) ? m.rows[(size_t)(off + j0 + t) * 12 + lane] : 0.0f;
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (j0 + t < len) {
                            cnt++;
                            if (lane < 9) acc += val[t];
                            else { const float count = (float)cnt; const float inv = 1 / count; acc = acc + inv * (val[t] - acc); }
                        }
                }
            }
            if (lane < 12) m.racc[(size_t)a * 12 + lane] = acc;
            float all[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) all[k] = __shfl(acc, k, 64);
            float rec[16];
            a_region_from_acc(all, cnt, rec);
            if (lane == 0) {
                for (int k = 0; k < 16; ++k) m.rrec[(size_t)a * 16 + k] = rec[k];
                m.rcnt[a] = cnt;
                m.lnext[m.rtail[a]] = m.rhead[b]; m.rtail[a] = m.rtail[b];
                m.ralive[b] = 0; m.parent[b] = a;
            }
        } else {
            // ---- edges that touch a or b
            for (uint32_t e = tid - 64; e < m.E; e += MG_THREADS - 64) {
                if (!m.ealive[e]) continue;
                const uint32_t p = m.ea[e], q = m.eb[e];
                const bool on_a = p == a || q == a, on_b = p == b || q == b;
                if (!on_a && !on_b) continue;
                const uint32_t x = (p == a || p == b) ? q : p;
                m.tl[atomicAdd(&s_ntouched, 1u)] = e;
                if (on_a) m.markA[x] = e + 1u; else m.markB[x] = e + 1u;
            }
        }
        __syncthreads();
        const uint32_t nt = s_ntouched;
        // ---- contains(): of (a,x) and (b,x) the entry that comes first in the old map survives
        for (uint32_t i = tid; i < nt; i += MG_THREADS) {
            const uint32_t e = m.tl[i];
            const uint32_t p = m.ea[e], q = m.eb[e];
            const bool on_a = p == a || q == a;
            const uint32_t x = (p == a || p == b) ? q : p;
            const uint32_t partner = on_a ? m.markB[x] : m.markA[x];
            if (partner && edge_before(m, partner - 1u, e)) m.ealive[e] = 0;
        }
        __syncthreads();
        // ---- re-weight the survivors (delta(), src/clustering.cpp:438-463)
        for (uint32_t i = tid; i < nt; i += MG_THREADS) {
            const uint32_t e = m.tl[i];
            const uint32_t p = m.ea[e], q = m.eb[e];
            const uint32_t x = (p == a || p == b) ? q : p;
            m.markA[x] = 0u; m.markB[x] = 0u;
            if (!m.ealive[e]) continue;
            const uint32_t lo = a < x ? a : x, hi = a < x ? x : a;
            int err = 0;
            const float w = a_edge_weight(mp, m.rrec + (size_t)lo * 16, m.rrec + (size_t)hi * 16, &err);
            if (err) m.dc->error = err;
            const uint32_t ev = atomicAdd(&s_nevents, 1u);
            if (ev >= m.ev_cap) { m.dc->error = F3DS_ERR_UNSUPPORTED; continue; }
            const uint32_t ku = n_weight_key(w);
            m.ev_epoch[ev] = epoch; m.ev_key[ev] = ku; m.ev_prev[ev] = m.ehist[e];
            m.ea[e] = lo; m.eb[e] = hi; m.ew[e] = w; m.eku[e] = ku; m.ehist[e] = (int)ev;
        }
        if (tid == 0) s_ntouched = 0;
        __syncthreads();
        if (m.dc->error) break;
    }
    __syncthreads();
    if (tid == 0) { m.dc->n_merges = s_nmerges; m.dc->n_events = s_nevents; }
}

// ------------------------------------------------------------------------------------------------
// stage 5, fast path: the same merge loop with the edge keys and endpoints resident in LDS.
//   * akey[e]   order key of the edge's weight (0xFFFFFFFF = edge gone); groups of 64 edges keep
//               their minimum (gkey/gidx), so the next merge is a two-level wave minimum instead of
//               a scan of all edges; only groups whose edges changed are recomputed.
//   * eab[e]    endpoints packed a<<16|b: "which edges touch a or b" is a pure LDS scan.
//   * a region's leaves (original supervoxels, in voxels_ concatenation order) are an array in a
//     pool, so region b's voxel rows can be gathered by the whole workgroup into an LDS staging
//     tile (block scan of the leaf lengths + binary search per row); wave 0 then continues a's
//     nine ordered sums and wave 1 a's running colour mean straight from LDS.
// Ties between equal keys fall back to the history comparator, exactly like k_merge.
// ------------------------------------------------------------------------------------------------
constexpr int ML_THREADS = 512;
constexpr uint32_t KEY_DEAD = 0xFFFFFFFFu;
struct MergeLds {
    uint32_t* pool; uint32_t pool_cap; uint32_t* rstart; uint32_t* rnleaf; uint32_t* rcap;
    uint32_t Ecap, G, caprows, stage_off, stop_key;
};
// wave-wide unsigned minimum with DPP lane swizzles (quad, half-row, row, row broadcasts): six VALU
// steps instead of six ds_bpermute round trips; every lane returns the result.
__device__ inline uint32_t wave_min_u32(uint32_t v) {
#define F3DS_DPP_MIN(ctrl, rmask)                                                                          \
    { const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xF, false); v = t_ < v ? t_ : v; }
    F3DS_DPP_MIN(0xB1, 0xF)     // quad_perm [1,0,3,2]
    F3DS_DPP_MIN(0x4E, 0xF)     // quad_perm [2,3,0,1]
    F3DS_DPP_MIN(0x141, 0xF)    // row_half_mirror
    F3DS_DPP_MIN(0x140, 0xF)    // row_mirror: every lane of a row holds the row minimum
    F3DS_DPP_MIN(0x142, 0xA)    // row_bcast15 into rows 1 and 3
    F3DS_DPP_MIN(0x143, 0xC)    // row_bcast31 into rows 2 and 3
#undef F3DS_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ inline bool edge_before_k(const MergeDev& m, uint32_t e, uint32_t ke, uint32_t f, uint32_t kf) {
    EdgeHist H{m.ev_epoch, m.ev_key, m.ev_prev};
    return a_edge_before(H, e, ke, m.ehist[e], f, kf, m.ehist[f]);
}
__global__ __launch_bounds__(ML_THREADS) void k_merge_lds(const MergeDev* frames_m, const MergeLds* frames_x) {
    // one workgroup per frame: a batch of frames merges concurrently inside a single dispatch
    const MergeDev m = frames_m[blockIdx.x];
    const MergeLds x = frames_x[blockIdx.x];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* akey = reinterpret_cast<uint32_t*>(smem);
    uint32_t* eab = akey + x.Ecap;
    uint32_t* gkey = eab + x.Ecap;
    uint32_t* gidx = gkey + x.G;
    uint32_t* lstart = gidx + x.G;                 // ML_THREADS + 1
    uint32_t* lsrc = lstart + ML_THREADS + 1;      // ML_THREADS
    unsigned char* gdirty = reinterpret_cast<unsigned char*>(lsrc + ML_THREADS);
    float* stage = reinterpret_cast<float*>(smem + x.stage_off);
    float* sinv = stage + (size_t)x.caprows * 12;   // 1/count of every staged row (the colour running mean)
    __shared__ uint32_t s_a, s_b, s_nt, s_nmerges, s_nevents, s_pool_end, s_best;
    __shared__ int s_stop;
    __shared__ float s_acc[12];
    __shared__ float s_rec[16];
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    constexpr int NW = ML_THREADS / 64;
    MergeParams mp = m.mp;
    if (mp.merging == 1) mp.lambda = m.dc->lambda;
    for (uint32_t e = tid; e < x.Ecap; e += ML_THREADS) {
        akey[e] = e < m.E ? m.eku[e] : KEY_DEAD;
        eab[e] = e < m.E ? ((m.ea[e] << 16) | m.eb[e]) : 0u;
    }
    for (uint32_t g = tid; g < x.G; g += ML_THREADS) gdirty[g] = 1;
    if (tid == 0) { s_nt = 0; s_nmerges = 0; s_nevents = m.E; s_stop = 0; s_pool_end = m.S0 + 1u; }
    __syncthreads();
#ifdef F3DS_MERGE_PROF   // per-phase shader-clock totals, printed by thread 0 (make PROF=1)
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define TPH(i) do { if (tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tph[i] += now_ - tlast; tlast = now_; } } while (0)
#else
#define TPH(i) do { } while (0)
#endif
    for (uint32_t epoch = 1;; ++epoch) {
        // ---- minima of the groups whose edges changed
        for (uint32_t g = wave; g < x.G; g += NW) {
            if (!gdirty[g]) continue;
            const uint32_t e = g * 64u + lane;
            const uint32_t k = akey[e];
            const uint32_t kmin = wave_min_u32(k);
            uint64_t cand = __ballot(k == kmin);
            uint32_t idx = g * 64u + (uint32_t)__builtin_ctzll(cand);
            if (kmin != KEY_DEAD && (cand & (cand - 1ull))) {          // equal keys: weight_map order decides
                cand &= cand - 1ull;
                while (cand) { const uint32_t f = g * 64u + (uint32_t)__builtin_ctzll(cand); cand &= cand - 1ull; if (edge_before_k(m, f, kmin, idx, kmin)) idx = f; }
            }
            if (lane == 0) { gkey[g] = kmin; gidx[g] = idx; gdirty[g] = 0; }
        }
        __syncthreads();
        TPH(0);
        // ---- next = *weight_map.begin()
        if (wave == 0) {
            uint32_t kloc = KEY_DEAD;
            for (uint32_t g = lane; g < x.G; g += 64u) { const uint32_t k = gkey[g]; kloc = k < kloc ? k : kloc; }
            const uint32_t kmin = wave_min_u32(kloc);
            int stop = !(kmin < x.stop_key);
            uint32_t best = 0;
            if (!stop) {
                uint32_t cnt = 0, first = 0;
                for (uint32_t g = lane; g < x.G; g += 64u) if (gkey[g] == kmin) { if (!cnt) first = g; cnt++; }
                const uint64_t has = __ballot(cnt > 0);
                uint32_t tot = cnt;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
                if (tot == 1u) best = gidx[__shfl(first, __builtin_ctzll(has), 64)];
                else {
                    bool have = false;
                    for (uint32_t g = 0; g < x.G; ++g)
                        if (gkey[g] == kmin) { const uint32_t f = gidx[g]; if (!have || edge_before_k(m, f, kmin, best, kmin)) { best = f; have = true; } }
                }
            }
            if (lane == 0) {
                s_stop = stop;
                if (!stop) {
                    s_best = best;
                    const uint32_t ab = eab[best];
                    s_a = ab >> 16; s_b = ab & 0xFFFFu;
                    const uint32_t k = s_nmerges++;
                    // weight bits from the order key (weights are never -0 or NaN here)
                    const uint32_t wb = (kmin & 0x80000000u) ? (kmin & 0x7fffffffu) : ~kmin;
                    m.merges[k * 3] = s_a; m.merges[k * 3 + 1] = s_b; m.merges[k * 3 + 2] = wb;
                    akey[best] = KEY_DEAD; gdirty[best >> 6] = 1;
                }
            }
        }
        __syncthreads();
        TPH(1);
        if (s_stop) break;
        const uint32_t a = s_a, b = s_b;
        // ---- edges that touch a or b (LDS scan), leaf array of the merged region
        for (uint32_t e = tid; e < x.Ecap; e += ML_THREADS) {
            if (akey[e] == KEY_DEAD) continue;
            const uint32_t ab = eab[e], p = ab >> 16, q = ab & 0xFFFFu;
            const bool on_a = p == a || q == a, on_b = p == b || q == b;
            if (!on_a && !on_b) continue;
            const uint32_t xx = (p == a || p == b) ? q : p;
            m.tl[atomicAdd(&s_nt, 1u)] = e;
            if (on_a) m.markA[xx] = e + 1u; else m.markB[xx] = e + 1u;
        }
        // leaf array of a ++ b: append in place while a's segment has room, else move to a segment twice the size
        const uint32_t a_start = x.rstart[a], na = x.rnleaf[a], cap_a = x.rcap[a], b_start = x.rstart[b], nb = x.rnleaf[b];
        const bool in_place = na + nb <= cap_a;
        const uint32_t new_start = in_place ? a_start : s_pool_end;
        const uint32_t new_cap = in_place ? cap_a : 2u * (na + nb);
        if (!in_place && new_start + new_cap > x.pool_cap) { if (tid == 0) m.dc->error = F3DS_ERR_UNSUPPORTED; break; }
        if (in_place) { for (uint32_t i = tid; i < nb; i += ML_THREADS) x.pool[a_start + na + i] = x.pool[b_start + i]; }
        else { for (uint32_t i = tid; i < na + nb; i += ML_THREADS) x.pool[new_start + i] = i < na ? x.pool[a_start + i] : x.pool[b_start + (i - na)]; }
        float acc = 0.0f;
        const uint32_t cnt_a = m.rcnt[a];
        if (wave == 0 && lane < 9) acc = m.racc[(size_t)a * 12 + lane];
        if (wave == 1 && lane < 3) acc = m.racc[(size_t)a * 12 + 9 + lane];
        uint32_t rows_done = 0;
        bool dedupe_done = false;
        __syncthreads();
        TPH(2);
        const uint32_t nt = s_nt;
        // ---- voxels_new = voxels_a ++ voxels_b: gather b's rows to LDS, continue a's ordered sums
        for (uint32_t lc = 0; lc < nb; lc += ML_THREADS) {
            const uint32_t nl = nb - lc < (uint32_t)ML_THREADS ? nb - lc : (uint32_t)ML_THREADS;
            uint32_t len = 0, off = 0;
            if ((uint32_t)tid < nl) { const uint32_t leaf = x.pool[b_start + lc + tid]; len = m.llen[leaf]; off = m.loff[leaf]; }
            uint32_t T;
            const uint32_t inc = block_incl_scan<ML_THREADS>(len, &T);
            lstart[tid] = inc - len; lsrc[tid] = off;
            if (tid == 0) lstart[ML_THREADS] = T;
            __syncthreads();
            for (uint32_t R0 = 0; R0 < T; R0 += x.caprows) {
                const uint32_t nr = T - R0 < x.caprows ? T - R0 : x.caprows;
                for (uint32_t r = tid; r < nr; r += ML_THREADS) {
                    const uint32_t rr = R0 + r;
                    uint32_t lo = 0, hi = nl;                     // last leaf with lstart <= rr
                    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (lstart[mid] <= rr) lo = mid; else hi = mid; }
                    const float4* src = reinterpret_cast<const float4*>(m.rows + (size_t)(lsrc[lo] + (rr - lstart[lo])) * 12);
                    float4* dst = reinterpret_cast<float4*>(stage + (size_t)r * 12);
                    const float4 q0 = src[0], q1 = src[1], q2 = src[2];
                    dst[0] = q0; dst[1] = q1; dst[2] = q2;
                    const float count = (float)(cnt_a + rows_done + r + 1u);
                    sinv[r] = 1 / count;
                }
                __syncthreads();
                if (wave == 0) {
                    if (lane < 9) {
#pragma unroll 8
                        for (uint32_t j = 0; j < nr; ++j) acc += stage[j * 12 + lane];
                    }
                } else if (wave == 1) {
                    if (lane < 3)
#pragma unroll 8
                        for (uint32_t j = 0; j < nr; ++j) acc = acc + sinv[j] * (stage[j * 12 + 9 + lane] - acc);
                } else if (!dedupe_done) {
                    // contains(): of (a,x) and (b,x) the entry that comes first in the old map survives
                    for (uint32_t i = tid - 128; i < nt; i += ML_THREADS - 128) {
                        const uint32_t e = m.tl[i];
                        const uint32_t ab = eab[e], p = ab >> 16, q = ab & 0xFFFFu;
                        const bool on_a = p == a || q == a;
                        const uint32_t xx = (p == a || p == b) ? q : p;
                        const uint32_t partner = on_a ? m.markB[xx] : m.markA[xx];
                        if (partner && edge_before_k(m, partner - 1u, akey[partner - 1u], e, akey[e])) m.tl[i] = e | 0x80000000u;
                    }
                }
                dedupe_done = true;
                rows_done += nr;
                __syncthreads();
            }
        }
        TPH(3);
        if (wave == 0 && lane < 9) { s_acc[lane] = acc; m.racc[(size_t)a * 12 + lane] = acc; }
        if (wave == 1 && lane < 3) { s_acc[9 + lane] = acc; m.racc[(size_t)a * 12 + 9 + lane] = acc; }
        __syncthreads();
        // ---- the merged region's record: wave 0 centroid + PCA normal, wave 1 mean colour -> Lab
        if (wave == 0) {
            float all[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) all[k] = s_acc[k];
            const uint32_t cnt = cnt_a + rows_done;
            const float c = (float)cnt;
            const float cen[3] = {all[6] / c, all[7] / c, all[8] / c};
            float n4[4];
            n_plane_normal(all, cnt, cen, n4);
            if (lane == 0) {
                for (int k = 0; k < 3; ++k) { s_rec[k] = cen[k]; s_rec[3 + k] = n4[k]; m.rrec[(size_t)a * 16 + k] = cen[k]; m.rrec[(size_t)a * 16 + 3 + k] = n4[k]; }
                m.rcnt[a] = cnt;
                x.rstart[a] = new_start; x.rnleaf[a] = na + nb; x.rcap[a] = new_cap; if (!in_place) s_pool_end = new_start + new_cap;
                m.ralive[b] = 0; m.parent[b] = a;
            }
        } else if (wave == 1) {
            // n_rgb2lab with the three gamma curves and the three cube roots evaluated in lanes 0..2
            const float mr = s_acc[9], mg = s_acc[10], mb = s_acc[11];
            const float mine = lane == 0 ? mr : (lane == 1 ? mg : mb);
            const float v = mine / 255;
            const float cl = v <= 0.04045f ? v / 12.92f : (float)m_pow_pos((double)((v + 0.055f) / 1.055f), 2.4);
            const float c0 = __shfl(cl, 0, 64), c1 = __shfl(cl, 1, 64), c2 = __shfl(cl, 2, 64);
            const float X = (c0 * 0.412453f + c1 * 0.357580f + c2 * 0.180423f) / 0.950456f;
            const float Y = (c0 * 0.212671f + c1 * 0.715160f + c2 * 0.072169f);
            const float Z = (c0 * 0.019334f + c1 * 0.119193f + c2 * 0.950227f) / 1.088754f;
            const float fl = n_lab_f(lane == 0 ? X : (lane == 1 ? Y : Z));
            const float fx = __shfl(fl, 0, 64), fy = __shfl(fl, 1, 64), fz = __shfl(fl, 2, 64);
            const float L = Y > 0.008856f ? 116.0f * fy - 16.0f : 903.3f * Y;
            if (lane == 0) {
                const float rec[6] = {mr, mg, mb, L, 500.0f * (fx - fy), 200.0f * (fy - fz)};
                for (int k = 0; k < 6; ++k) { s_rec[6 + k] = rec[k]; m.rrec[(size_t)a * 16 + 6 + k] = rec[k]; }
            }
        }
        __syncthreads();
        TPH(4);
        // ---- re-weight the surviving incident edges (delta(), src/clustering.cpp:438-463)
        for (uint32_t i = tid; i < nt; i += ML_THREADS) {
            const uint32_t te = m.tl[i], e = te & 0x7fffffffu;
            const uint32_t ab = eab[e], p = ab >> 16, q = ab & 0xFFFFu;
            const uint32_t xx = (p == a || p == b) ? q : p;
            m.markA[xx] = 0u; m.markB[xx] = 0u;
            gdirty[e >> 6] = 1;
            if (te & 0x80000000u) { akey[e] = KEY_DEAD; continue; }
            const uint32_t lo = a < xx ? a : xx, hi = a < xx ? xx : a;
            float r1[12], r2[12];          // record of the lower label first, like delta(segments.at(first), segments.at(second))
            {
                const float4* gx = reinterpret_cast<const float4*>(m.rrec + (size_t)xx * 16);
                const float4 x0 = gx[0], x1 = gx[1], x2 = gx[2];
                const float rx[12] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
                const bool a_first = a < xx;
#pragma unroll
                for (int k = 0; k < 12; ++k) { const float ra = s_rec[k]; r1[k] = a_first ? ra : rx[k]; r2[k] = a_first ? rx[k] : ra; }
            }
            int err = 0;
            const float w = a_edge_weight(mp, r1, r2, &err);
            if (err) m.dc->error = err;
            const uint32_t ev = atomicAdd(&s_nevents, 1u);
            if (ev >= m.ev_cap) { m.dc->error = F3DS_ERR_UNSUPPORTED; continue; }
            const uint32_t ku = n_weight_key(w);
            m.ev_epoch[ev] = epoch; m.ev_key[ev] = ku; m.ev_prev[ev] = m.ehist[e];
            m.ehist[e] = (int)ev;
            akey[e] = ku; eab[e] = (lo << 16) | hi;
        }
        if (tid == 0) s_nt = 0;
        __syncthreads();
        TPH(5);
        if (m.dc->error) break;
    }
    __syncthreads();
    if (tid == 0) { m.dc->n_merges = s_nmerges; m.dc->n_events = s_nevents; }
#ifdef F3DS_MERGE_PROF
    if (tid == 0) printf("k_merge_lds cycles/merge: groupmin %.0f argmin %.0f touched+pool %.0f gather+fold %.0f record %.0f weights %.0f (merges %u)\n",
                         (double)tph[0] / s_nmerges, (double)tph[1] / s_nmerges, (double)tph[2] / s_nmerges, (double)tph[3] / s_nmerges, (double)tph[4] / s_nmerges,
                         (double)tph[5] / s_nmerges, s_nmerges);
#endif
#undef TPH
}

// ------------------------------------------------------------------------------------------------
// stage 6: region ids and per-point labels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_roots(uint32_t S0, const uint32_t* parent, const unsigned char* ralive, uint32_t* root, uint32_t* flags) {
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h <= S0; h += gridDim.x * blockDim.x) {
        uint32_t r = h;
        while (parent[r] != r) r = parent[r];
        root[h] = r;
        flags[h] = (h > 0 && ralive[h]) ? 1u : 0u;
    }
}
__global__ __launch_bounds__(256) void k_point_labels(uint32_t n, const int* pt_voxel, const uint32_t* owner, const uint32_t* root, const uint32_t* incl,
                                                     uint32_t S0, uint32_t* labels, DevCounters* dc) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int v = pt_voxel[i];
        uint32_t l = F3DS_NO_LABEL;
        if (v >= 0) { const uint32_t o = owner[v]; if (o) l = incl[root[o]] - 1u; }
        labels[i] = l;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) dc->n_regions = incl[S0];
}
}  // namespace


// ================================================================================================
// host side
// ================================================================================================
struct f3ds_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[9] = {};
    DevCounters* d_dc = nullptr;
    DevCounters* h_dc = nullptr;       // pinned
    GridInfo* d_grid = nullptr;
    GridInfo* h_grid = nullptr;        // pinned
    SeedGrid* d_sgrid = nullptr;
    SeedGrid* h_sgrid = nullptr;       // pinned
    // frame state
    bool have_frame = false;
    f3ds_params prm;
    FrameArgs fa;
    uint32_t n = 0, V = 0, C = 0, S0 = 0, E = 0, hmask = 0;
    f3ds_result res;
    // device scratch (grow-only)
    Buf pts, keys0, keys1, vals0, vals1, flags, incl, tiles, hist, seg_start, pt_voxel, labels;
    Buf vkey, vcount, vf, nbr, hkeys, hvals, boxes, ckey, cell_start, chk, chv, seed_orig, keep, seed_kept;
    Buf owner0, owner1, dist0, dist1, R, hc, hcount, hlo, hhi, ghost_vox, ghost_active, ghost_done, ghost_head, ghost_next;
    Buf loff, rows, row_voxel, racc0, rcnt0, rrec0, ralive0, ehk, ekeys0, ekeys1, evals0, evals1, ea0, eb0;
    Buf ea, eb, ew, eku, ehist, ealive, ev_epoch, ev_key, ev_prev, racc, rcnt, rrec, ralive, rhead, rtail, lnext, parent, markA, markB, tl, merges;
    Buf deltas, skeys0, skeys1, svals0, svals1, cdf_hist, cdf, root, rflags, pool, rstart, rnleaf, rcap, rincl;
    bool merge_in_lds = false;
    MergeDev mdev; MergeLds mlds; uint32_t merge_dyn = 0; float host_lambda = 0.5f;
    Buf batch_args, sweep_args, nbrT, ownR;
};

namespace {

template <class T>
int ensure(Buf& b, size_t count, T** out) {
    size_t bytes = count * sizeof(T);
    if (bytes < 256) bytes = 256;
    if (b.cap < bytes) {
        if (b.p) { HIPCHECK(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
        size_t want = bytes + bytes / 4;
        HIPCHECK(hipMalloc(&b.p, want));
        b.cap = want;
    }
    *out = reinterpret_cast<T*>(b.p);
    return F3DS_OK;
}
#define ENSURE(buf, T, count, ptr) do { int rc_ = ensure<T>(buf, (size_t)(count), &ptr); if (rc_) return rc_; } while (0)

inline uint32_t grid_for(size_t work, int block) {
    size_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;          // grid-stride beyond 256 CUs x 8 workgroups
    return (uint32_t)g;
}
inline uint32_t pow2_ge(size_t x) { uint32_t p = 1; while (p < x) p <<= 1; return p; }

// inclusive scan of n uint32 values (in -> out); tiles is scratch
int scan_u32(f3ds_ctx* c, const uint32_t* in, uint32_t* out, uint32_t n) {
    if (n == 0) return F3DS_OK;
    const uint32_t nt = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t* tiles;
    ENSURE(c->tiles, uint32_t, nt, tiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3(nt), dim3(SCAN_THREADS), 0, c->stream, in, out, tiles, n);
    if (nt > 1) {
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, c->stream, tiles, nt);
        hipLaunchKernelGGL(k_scan_add, dim3(nt), dim3(SCAN_THREADS), 0, c->stream, out, tiles, n);
    }
    return F3DS_OK;
}
// stable sort of (key,val) pairs on the low `total_bits` bits; the result ends in *keys_out/*vals_out
int radix_sort(f3ds_ctx* c, uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, uint32_t n, int total_bits, uint64_t** keys_out, uint32_t** vals_out) {
    *keys_out = k0; *vals_out = v0;
    if (n == 0 || total_bits <= 0) return F3DS_OK;
    const int passes = (total_bits + 7) / 8;
    const int per = (total_bits + passes - 1) / passes;
    const uint32_t nb = (n + RS_TILE - 1) / RS_TILE;
    uint32_t* hist;
    ENSURE(c->hist, uint32_t, (size_t)256 * nb, hist);
    int shift = 0;
    for (int p = 0; p < passes; ++p) {
        const int bits = (total_bits - shift) < per ? (total_bits - shift) : per;
        hipLaunchKernelGGL(k_radix_hist, dim3(nb), dim3(RS_THREADS), 0, c->stream, (const uint64_t*)k0, n, shift, bits, hist, nb);
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, c->stream, hist, (uint32_t)((1u << bits) * nb));
        hipLaunchKernelGGL(k_radix_scatter, dim3(nb), dim3(RS_THREADS), 0, c->stream, (const uint64_t*)k0, (const uint32_t*)v0, k1, v1, n, shift, bits,
                           (const uint32_t*)hist, nb);
        std::swap(k0, k1); std::swap(v0, v1);
        shift += bits;
    }
    *keys_out = k0; *vals_out = v0;
    return F3DS_OK;
}
int sync_counters(f3ds_ctx* c) {
    HIPCHECK(hipMemcpyAsync(c->h_dc, c->d_dc, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return F3DS_OK;
}
int bits_for(uint64_t max_value) { int b = 0; while (b < 64 && (max_value >> b)) ++b; return b; }

int finish_empty(f3ds_ctx* c, uint32_t* point_labels, int labels_on_device, f3ds_result* result) {
    uint32_t* d_labels;
    ENSURE(c->labels, uint32_t, c->n ? c->n : 1, d_labels);
    if (c->n) {
        HIPCHECK(hipMemsetAsync(d_labels, 0xFF, (size_t)c->n * 4, c->stream));
        if (point_labels) HIPCHECK(hipMemcpyAsync(point_labels, d_labels, (size_t)c->n * 4, labels_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHECK(hipStreamSynchronize(c->stream));
    c->have_frame = false;
    if (result) *result = c->res;
    return F3DS_OK;
}

// stages 4b..6: Clustering::cluster(threshold) on the supervoxels held by the context
// stages 4b: everything of Clustering::cluster(threshold) up to (not including) the merge loop
int run_cluster_front(f3ds_ctx* c, const f3ds_params* prm) {
    hipStream_t st = c->stream;
    const uint32_t S0 = c->S0, E = c->E;
    // main(): set_merging / set_lambda / set_bins_num (src/supervoxel_clustering.cpp:415-423)
    float lambda = 0.5f; int bins = 500;
    if (prm->merging == F3DS_MANUAL_LAMBDA && prm->lambda != 0) { if (prm->lambda < 0 || prm->lambda > 1) return F3DS_ERR_RANGE; lambda = prm->lambda; }
    if (prm->merging == F3DS_EQUALIZATION && prm->bins != 0) { if (prm->bins < 0) return F3DS_ERR_RANGE; bins = (short)prm->bins; }
    if (prm->merging < 0 || prm->merging > 2 || prm->color_metric < 0 || prm->color_metric > 1 || prm->geom_metric < 0 || prm->geom_metric > 1) return F3DS_ERR_ARG;
    MergeDev m;
    memset(&m, 0, sizeof m);
    m.E = E; m.S0 = S0; m.threshold = prm->threshold; m.dc = c->d_dc;
    m.ev_cap = E * 64u + 4096u;
    ENSURE(c->ea, uint32_t, E, m.ea); ENSURE(c->eb, uint32_t, E, m.eb); ENSURE(c->ew, float, E, m.ew); ENSURE(c->eku, uint32_t, E, m.eku);
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
    HIPCHECK(hipMemcpyAsync(m.racc, c->racc0.p, (size_t)(S0 + 1) * 12 * 4, hipMemcpyDeviceToDevice, st));
    HIPCHECK(hipMemcpyAsync(m.rrec, c->rrec0.p, (size_t)(S0 + 1) * 16 * 4, hipMemcpyDeviceToDevice, st));
    HIPCHECK(hipMemcpyAsync(m.rcnt, c->rcnt0.p, (size_t)(S0 + 1) * 4, hipMemcpyDeviceToDevice, st));
    HIPCHECK(hipMemcpyAsync(m.ralive, c->ralive0.p, (size_t)(S0 + 1), hipMemcpyDeviceToDevice, st));
    if (E) {
        HIPCHECK(hipMemcpyAsync(m.ea, c->ea0.p, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
        HIPCHECK(hipMemcpyAsync(m.eb, c->eb0.p, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
    }
    // fast path: keys + endpoints of every edge and a row staging tile fit the 160 KB LDS of one CU
    MergeLds xl;
    memset(&xl, 0, sizeof xl);
    xl.Ecap = (E + 63u) & ~63u; xl.G = xl.Ecap / 64u;
    const uint32_t lds_fixed = xl.Ecap * 8u + xl.G * 8u + (2u * ML_THREADS + 1u) * 4u + ((xl.G + 15u) & ~15u);
    xl.stage_off = (lds_fixed + 15u) & ~15u;
    const uint32_t lds_budget = 160u * 1024u - 4096u;
    bool use_lds = E > 0 && S0 <= 65535u && xl.stage_off + 128u * 52u <= lds_budget && !getenv("F3DS_FORCE_GLOBAL_MERGE");
    if (use_lds) { xl.caprows = (lds_budget - xl.stage_off) / 52u; if (xl.caprows > 2048u) xl.caprows = 2048u; }
    uint32_t logS = 1; while ((1u << logS) < S0 + 2u) ++logS;
    xl.pool_cap = (S0 + 1u) * (4u * logS + 8u);
    ENSURE(c->pool, uint32_t, xl.pool_cap, xl.pool); ENSURE(c->rstart, uint32_t, S0 + 1, xl.rstart); ENSURE(c->rnleaf, uint32_t, S0 + 1, xl.rnleaf);
    ENSURE(c->rcap, uint32_t, S0 + 1, xl.rcap);
    xl.stop_key = (prm->threshold != prm->threshold) ? 0u : n_weight_key(prm->threshold);
    c->merge_in_lds = use_lds;
    hipLaunchKernelGGL(k_region_reset, dim3(grid_for(S0 + 1, 256)), dim3(256), 0, st, S0, (const uint32_t*)c->hcount.p, m.rhead, m.rtail, m.lnext, m.parent, m.markA, m.markB,
                       xl.pool, xl.rstart, xl.rnleaf, xl.rcap);
    float* cdf = nullptr;
    m.mp.color_metric = prm->color_metric; m.mp.geom_metric = prm->geom_metric; m.mp.merging = prm->merging; m.mp.lambda = lambda; m.mp.bins = bins;
    if (E) {
        uint64_t *sk0 = nullptr, *sk1 = nullptr; uint32_t *sv0 = nullptr, *sv1 = nullptr;
        if (prm->merging == F3DS_ADAPTIVE_LAMBDA) {
            ENSURE(c->skeys0, uint64_t, (size_t)E * 2, sk0); ENSURE(c->skeys1, uint64_t, (size_t)E * 2, sk1);
            ENSURE(c->svals0, uint32_t, (size_t)E * 2, sv0); ENSURE(c->svals1, uint32_t, (size_t)E * 2, sv1);
        }
        hipLaunchKernelGGL(k_edge_deltas, dim3(grid_for(E, 256)), dim3(256), 0, st, E, (const uint32_t*)m.ea, (const uint32_t*)m.eb, (const float*)m.rrec,
                           prm->color_metric, prm->geom_metric, deltas, sk0, sv0);
        if (prm->merging == F3DS_ADAPTIVE_LAMBDA) {
            uint64_t* ks; uint32_t* vs;
            int rc = radix_sort(c, sk0, sv0, sk1, sv1, E * 2u, 33, &ks, &vs);
            if (rc) return rc;
            hipLaunchKernelGGL(k_lambda, dim3(1), dim3(64), 0, st, E, (const float*)deltas, (const uint32_t*)vs, c->d_dc);
        } else if (prm->merging == F3DS_EQUALIZATION) {
            uint32_t* hist; ENSURE(c->cdf_hist, uint32_t, (size_t)2 * (bins > 0 ? bins : 1), hist);
            ENSURE(c->cdf, float, (size_t)2 * (bins > 0 ? bins : 1), cdf);
            HIPCHECK(hipMemsetAsync(hist, 0, (size_t)2 * (bins > 0 ? bins : 1) * 4, st));
            hipLaunchKernelGGL(k_cdf_hist, dim3(grid_for((size_t)E * 2, 256)), dim3(256), 0, st, E, (const float*)deltas, bins, hist, c->d_dc);
            hipLaunchKernelGGL(k_cdf_scan, dim3(1), dim3(64), 0, st, E, bins, (const uint32_t*)hist, cdf);
            m.mp.cdf_c = cdf; m.mp.cdf_g = cdf + bins;
        }
        hipLaunchKernelGGL(k_edge_weights, dim3(grid_for(E, 256)), dim3(256), 0, st, m, (const float*)deltas);
    }
    c->mdev = m; c->mlds = xl; c->merge_dyn = use_lds ? xl.stage_off + xl.caprows * 52u : 0u; c->host_lambda = lambda;
    return F3DS_OK;
}
// stage 5 for a set of frames whose fronts are complete: the LDS-resident merge loops of all of them
// run as ONE dispatch (one workgroup per frame) on `st`; frames that do not fit LDS get k_merge each.
int merge_launch(f3ds_ctx** cs, int nctx, hipStream_t st) {
    f3ds_ctx* c0 = cs[0];
    std::vector<MergeDev> ms; std::vector<MergeLds> xs; uint32_t dyn = 0;
    for (int i = 0; i < nctx; ++i) {
        HIPCHECK(hipEventRecord(cs[i]->ev[5], st));
        if (cs[i]->merge_in_lds) { ms.push_back(cs[i]->mdev); xs.push_back(cs[i]->mlds); if (cs[i]->merge_dyn > dyn) dyn = cs[i]->merge_dyn; }
    }
    if (!ms.empty()) {
        unsigned char* args;
        const size_t bytes_m = ms.size() * sizeof(MergeDev), bytes_x = xs.size() * sizeof(MergeLds);
        ENSURE(c0->batch_args, unsigned char, bytes_m + bytes_x, args);
        HIPCHECK(hipMemcpyAsync(args, ms.data(), bytes_m, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(args + bytes_m, xs.data(), bytes_x, hipMemcpyHostToDevice, st));
        HIPCHECK(hipStreamSynchronize(st));        // the host vectors go out of scope below
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(k_merge_lds, dim3((uint32_t)ms.size()), dim3(ML_THREADS), dyn, st, (const MergeDev*)args, (const MergeLds*)(args + bytes_m));
    }
    for (int i = 0; i < nctx; ++i)
        if (!cs[i]->merge_in_lds) hipLaunchKernelGGL(k_merge, dim3(1), dim3(MG_THREADS), 0, st, cs[i]->mdev);
    for (int i = 0; i < nctx; ++i) HIPCHECK(hipEventRecord(cs[i]->ev[6], st));
    return F3DS_OK;
}
// stage 6: region ids and per-point labels
int run_cluster_tail(f3ds_ctx* c, const f3ds_params* prm, uint32_t* point_labels, int labels_on_device) {
    hipStream_t st = c->stream;
    const uint32_t S0 = c->S0, E = c->E, n = c->n;
    const MergeDev& m = c->mdev;
    const float lambda = c->host_lambda;
    uint32_t *root, *rflags, *rincl, *d_labels;
    ENSURE(c->root, uint32_t, S0 + 1, root); ENSURE(c->rflags, uint32_t, S0 + 1, rflags); ENSURE(c->rincl, uint32_t, S0 + 1, rincl);
    ENSURE(c->labels, uint32_t, n, d_labels);
    hipLaunchKernelGGL(k_roots, dim3(grid_for(S0 + 1, 256)), dim3(256), 0, st, S0, (const uint32_t*)m.parent, (const unsigned char*)m.ralive, root, rflags);
    { int rc = scan_u32(c, rflags, rincl, S0 + 1); if (rc) return rc; }
    hipLaunchKernelGGL(k_point_labels, dim3(grid_for(n, 256)), dim3(256), 0, st, n, (const int*)c->pt_voxel.p, (const uint32_t*)c->owner0.p, (const uint32_t*)root,
                       (const uint32_t*)rincl, S0, d_labels, c->d_dc);
    HIPCHECK(hipEventRecord(c->ev[7], st));
    if (point_labels) HIPCHECK(hipMemcpyAsync(point_labels, d_labels, (size_t)n * 4, labels_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    { int rc = sync_counters(c); if (rc) return rc; }
    HIPCHECK(hipGetLastError());
    if (c->h_dc->error) return c->h_dc->error;
    c->res.n_merges = c->h_dc->n_merges; c->res.n_regions = c->h_dc->n_regions;
    c->res.lambda = prm->merging == F3DS_ADAPTIVE_LAMBDA ? (E ? c->h_dc->lambda : __builtin_nanf("")) : lambda;
    c->prm.color_metric = prm->color_metric; c->prm.geom_metric = prm->geom_metric; c->prm.merging = prm->merging;
    c->prm.lambda = prm->lambda; c->prm.bins = prm->bins; c->prm.threshold = prm->threshold;
    return F3DS_OK;
}
int run_cluster(f3ds_ctx* c, const f3ds_params* prm, uint32_t* point_labels, int labels_on_device) {
    int rc = run_cluster_front(c, prm);
    if (rc) return rc;
    if ((rc = merge_launch(&c, 1, c->stream))) return rc;
    return run_cluster_tail(c, prm, point_labels, labels_on_device);
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
    HIPCHECK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    for (auto& e : c->ev) HIPCHECK(hipEventCreate(&e));
    HIPCHECK(hipMalloc((void**)&c->d_dc, sizeof(DevCounters)));
    HIPCHECK(hipHostMalloc((void**)&c->h_dc, sizeof(DevCounters), hipHostMallocDefault));
    HIPCHECK(hipMalloc((void**)&c->d_grid, sizeof(GridInfo)));
    HIPCHECK(hipHostMalloc((void**)&c->h_grid, sizeof(GridInfo), hipHostMallocDefault));
    HIPCHECK(hipMalloc((void**)&c->d_sgrid, sizeof(SeedGrid)));
    HIPCHECK(hipHostMalloc((void**)&c->h_sgrid, sizeof(SeedGrid), hipHostMallocDefault));
    memset(&c->res, 0, sizeof c->res);
    *out = c;
    return F3DS_OK;
}

void f3ds_destroy(f3ds_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    Buf* bufs = &c->pts;
    const size_t nb = (reinterpret_cast<char*>(&c->rincl) - reinterpret_cast<char*>(&c->pts)) / sizeof(Buf) + 1;
    for (size_t i = 0; i < nb; ++i) if (bufs[i].p) (void)hipFree(bufs[i].p);
    if (c->d_dc) (void)hipFree(c->d_dc);
    if (c->h_dc) (void)hipHostFree(c->h_dc);
    if (c->d_grid) (void)hipFree(c->d_grid);
    if (c->h_grid) (void)hipHostFree(c->h_grid);
    if (c->d_sgrid) (void)hipFree(c->d_sgrid);
    if (c->h_sgrid) (void)hipHostFree(c->h_sgrid);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int f3ds_set_stream(f3ds_ctx* c, void* hip_stream) {
    if (!c) return F3DS_ERR_ARG;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return F3DS_OK;
}

}  // extern "C"

namespace {
// stages 0..4: returns 1 when the frame ended early (no voxels: labels are already written), 0 when
// the merge stage is prepared (c->mdev / c->mlds), < 0 on error
int segment_front_a(f3ds_ctx* c, const void* points, size_t n_, int points_on_device, const f3ds_params* prm, uint32_t* point_labels,
                    int labels_on_device, f3ds_result* result) {
    if (!c || !prm || (!points && n_) || n_ > 0x7fffffffull) return F3DS_ERR_ARG;
    if (!(prm->voxel_res > 0) || !(prm->seed_res > 0)) return F3DS_ERR_ARG;
    HIPCHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const uint32_t n = (uint32_t)n_;
    c->have_frame = false;
    c->prm = *prm; c->n = n; c->V = c->C = c->S0 = c->E = 0;
    memset(&c->res, 0, sizeof c->res);
    c->res.n_points = n;
    FrameArgs fa{prm->use_transform, prm->fold_negative_z, prm->leaf_order, prm->voxel_res, prm->seed_res, prm->w_color, prm->w_spatial, prm->w_normal};
    c->fa = fa;
    const int max_depth = (int)(1.8f * prm->seed_res / prm->voxel_res);      // [PCL-recall] SupervoxelClustering::extract
    const uint32_t sweeps = max_depth > 1 ? (uint32_t)(max_depth - 1) : 0u;
    c->res.sweeps = sweeps;

    // ---- stage 0: voxelise
    HIPCHECK(hipEventRecord(c->ev[0], st));
    const P16* d_pts;
    if (points_on_device) d_pts = (const P16*)points;
    else {
        P16* up; ENSURE(c->pts, P16, n ? n : 1, up);
        if (n) HIPCHECK(hipMemcpyAsync(up, points, (size_t)n * 16, hipMemcpyHostToDevice, st));
        d_pts = up;
    }
    {
        DevCounters init; memset(&init, 0, sizeof init);
        init.bbox[0] = init.bbox[1] = init.bbox[2] = 0xFFFFFFFFu;
        *c->h_dc = init;
        HIPCHECK(hipMemcpyAsync(c->d_dc, c->h_dc, sizeof init, hipMemcpyHostToDevice, st));
    }
    if (n) hipLaunchKernelGGL(k_bbox, dim3(grid_for(n, 256 * 8) < 512 ? grid_for(n, 256 * 8) : 512), dim3(256), 0, st, d_pts, n, fa, c->d_dc);
    hipLaunchKernelGGL(k_grid, dim3(1), dim3(1), 0, st, c->d_dc, prm->voxel_res, c->d_grid);
    HIPCHECK(hipMemcpyAsync(c->h_grid, c->d_grid, sizeof(GridInfo), hipMemcpyDeviceToHost, st));
    { int rc = sync_counters(c); if (rc) return rc; }
    c->res.n_finite = c->h_dc->n_finite;
    if (c->h_grid->error) return c->h_grid->error;
    c->res.octree_depth = (uint32_t)c->h_grid->depth;
    if (c->h_grid->empty || n == 0) { int rc_ = finish_empty(c, point_labels, labels_on_device, result); return rc_ ? rc_ : 1; }
    const int depth = c->h_grid->depth;
    uint64_t *k0, *k1, *ks; uint32_t *v0, *v1, *vs;
    ENSURE(c->keys0, uint64_t, n, k0); ENSURE(c->keys1, uint64_t, n, k1); ENSURE(c->vals0, uint32_t, n, v0); ENSURE(c->vals1, uint32_t, n, v1);
    hipLaunchKernelGGL(k_keys, dim3(grid_for(n, 256)), dim3(256), 0, st, d_pts, n, fa, (const GridInfo*)c->d_grid, k0, v0);
    { int rc = radix_sort(c, k0, v0, k1, v1, n, 3 * depth + 1, &ks, &vs); if (rc) return rc; }
    uint32_t *flags, *incl, *seg_start; int* pt_voxel;
    ENSURE(c->flags, uint32_t, n, flags); ENSURE(c->incl, uint32_t, n, incl); ENSURE(c->seg_start, uint32_t, (size_t)n + 1, seg_start);
    ENSURE(c->pt_voxel, int, n, pt_voxel);
    const uint64_t invalid = 1ull << (3 * depth);
    hipLaunchKernelGGL(k_heads, dim3(grid_for(n, 256)), dim3(256), 0, st, (const uint64_t*)ks, n, invalid, flags);
    { int rc = scan_u32(c, flags, incl, n); if (rc) return rc; }
    hipLaunchKernelGGL(k_segstart, dim3(grid_for(n, 256)), dim3(256), 0, st, (const uint64_t*)ks, (const uint32_t*)flags, (const uint32_t*)incl, n, invalid, seg_start,
                       &c->d_dc->n_voxels, &c->d_dc->n_valid);
    { int rc = sync_counters(c); if (rc) return rc; }
    const uint32_t V = c->h_dc->n_voxels;
    c->V = V; c->res.n_voxels = V;
    if (V == 0) { int rc_ = finish_empty(c, point_labels, labels_on_device, result); return rc_ ? rc_ : 1; }
    uint32_t *vkey, *vcount, *hvals; float* vf; int* nbr; uint64_t* hkeys;
    const uint32_t hcap = pow2_ge((size_t)V * 2 + 16);
    c->hmask = hcap - 1;
    ENSURE(c->vkey, uint32_t, (size_t)V * 3, vkey); ENSURE(c->vcount, uint32_t, V, vcount); ENSURE(c->vf, float, (size_t)V * 12, vf);
    ENSURE(c->nbr, int, (size_t)V * 27, nbr); ENSURE(c->hkeys, uint64_t, hcap, hkeys); ENSURE(c->hvals, uint32_t, hcap, hvals);
    int* nbrT; ENSURE(c->nbrT, int, (size_t)V * 27, nbrT);
    HIPCHECK(hipMemsetAsync(pt_voxel, 0xFF, (size_t)n * 4, st));
    HIPCHECK(hipMemsetAsync(hkeys, 0xFF, (size_t)hcap * 8, st));
    hipLaunchKernelGGL(k_voxel_accum, dim3(grid_for(V, 256)), dim3(256), 0, st, d_pts, (const uint64_t*)ks, (const uint32_t*)vs, (const uint32_t*)seg_start,
                       (const DevCounters*)c->d_dc, fa, (const GridInfo*)c->d_grid, vkey, vcount, vf, pt_voxel, hkeys, hvals, c->hmask);
    HIPCHECK(hipEventRecord(c->ev[1], st));
    // ---- stage 1: neighbours + normals
    hipLaunchKernelGGL(k_neighbors, dim3(grid_for((size_t)V * 27, 256)), dim3(256), 0, st, (const uint32_t*)vkey, (const DevCounters*)c->d_dc, (const GridInfo*)c->d_grid,
                       (const uint64_t*)hkeys, (const uint32_t*)hvals, c->hmask, nbr, nbrT);
    hipLaunchKernelGGL(k_normals, dim3(grid_for(V, 256)), dim3(256), 0, st, vf, (const int*)nbr, (const DevCounters*)c->d_dc);
    HIPCHECK(hipEventRecord(c->ev[2], st));
    // ---- stage 2: seeds
    const uint32_t nchunks = (V + SEED_CHUNK - 1) / SEED_CHUNK;
    float* boxes; ENSURE(c->boxes, float, (size_t)nchunks * 6, boxes);
    hipLaunchKernelGGL(k_chunkbox, dim3(nchunks), dim3(SEED_CHUNK), 0, st, (const float*)vf, (const DevCounters*)c->d_dc, boxes);
    hipLaunchKernelGGL(k_seed_grow, dim3(1), dim3(1024), 0, st, (const float*)vf, (const float*)boxes, c->d_dc, prm->seed_res, c->d_sgrid);
    HIPCHECK(hipMemcpyAsync(c->h_sgrid, c->d_sgrid, sizeof(SeedGrid), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    if (c->h_sgrid->error) return c->h_sgrid->error;
    const int sdepth = c->h_sgrid->depth;
    uint32_t* ckey; ENSURE(c->ckey, uint32_t, (size_t)V * 3, ckey);
    // (the point-sort buffers are free again: n >= V)
    hipLaunchKernelGGL(k_seed_keys, dim3(grid_for(V, 256)), dim3(256), 0, st, (const float*)vf, (const DevCounters*)c->d_dc, (const SeedGrid*)c->d_sgrid, ckey, k0, v0);
    uint64_t* cks; uint32_t* cvs;
    { int rc = radix_sort(c, k0, v0, k1, v1, V, 3 * sdepth, &cks, &cvs); if (rc) return rc; }
    uint32_t* cell_start; ENSURE(c->cell_start, uint32_t, (size_t)V + 1, cell_start);
    const uint64_t climit = sdepth >= 21 ? 0xFFFFFFFFFFFFFFFFull : (1ull << (3 * sdepth));
    hipLaunchKernelGGL(k_heads, dim3(grid_for(V, 256)), dim3(256), 0, st, (const uint64_t*)cks, V, climit, flags);
    { int rc = scan_u32(c, flags, incl, V); if (rc) return rc; }
    uint32_t* dummy_valid = &c->d_dc->seg_count;
    hipLaunchKernelGGL(k_segstart, dim3(grid_for(V, 256)), dim3(256), 0, st, (const uint64_t*)cks, (const uint32_t*)flags, (const uint32_t*)incl, V, climit, cell_start,
                       &c->d_dc->n_cells, dummy_valid);
    { int rc = sync_counters(c); if (rc) return rc; }
    const uint32_t C = c->h_dc->n_cells;
    c->C = C; c->res.n_seed_cells = C;
    // the sorted voxel list must survive further sorts: keep a copy
    uint32_t* sorted_vox; ENSURE(c->chv, uint32_t, V, sorted_vox);
    HIPCHECK(hipMemcpyAsync(sorted_vox, cvs, (size_t)V * 4, hipMemcpyDeviceToDevice, st));
    const uint32_t ccap = pow2_ge((size_t)C * 2 + 16);
    uint64_t* chk; uint32_t* chvals; int *seed_orig, *seed_kept; uint32_t* keep;
    ENSURE(c->chk, uint64_t, ccap, chk); ENSURE(c->ehk, uint32_t, ccap, chvals);   // ehk is reused below for the edge set
    ENSURE(c->seed_orig, int, C, seed_orig); ENSURE(c->seed_kept, int, C, seed_kept); ENSURE(c->keep, uint32_t, C, keep);
    HIPCHECK(hipMemsetAsync(chk, 0xFF, (size_t)ccap * 8, st));
    hipLaunchKernelGGL(k_cell_hash, dim3(grid_for(C, 256)), dim3(256), 0, st, (const uint32_t*)ckey, (const uint32_t*)sorted_vox, (const uint32_t*)cell_start,
                       (const DevCounters*)c->d_dc, chk, chvals, ccap - 1);
    hipLaunchKernelGGL(k_seed_nn, dim3(C), dim3(64), 0, st, (const float*)vf, (const uint32_t*)ckey, (const uint32_t*)sorted_vox, (const uint32_t*)cell_start,
                       (const DevCounters*)c->d_dc, (const SeedGrid*)c->d_sgrid, (const uint64_t*)chk, (const uint32_t*)chvals, ccap - 1, seed_orig);
    hipLaunchKernelGGL(k_seed_filter, dim3(C), dim3(64), 0, st, (const float*)vf, (const uint32_t*)ckey, (const uint32_t*)sorted_vox, (const uint32_t*)cell_start,
                       (const DevCounters*)c->d_dc, (const uint64_t*)chk, (const uint32_t*)chvals, ccap - 1, (const int*)seed_orig, a_radius_sq(prm->seed_res),
                       a_min_points(prm->seed_res, prm->voxel_res), keep);
    { int rc = scan_u32(c, keep, incl, C); if (rc) return rc; }
    hipLaunchKernelGGL(k_seed_compact, dim3(grid_for(C, 256)), dim3(256), 0, st, (const int*)seed_orig, (const uint32_t*)keep, (const uint32_t*)incl, c->d_dc, seed_kept);
    { int rc = sync_counters(c); if (rc) return rc; }
    const uint32_t S0 = c->h_dc->n_seeds;
    c->S0 = S0; c->res.n_seeds = S0;
    HIPCHECK(hipEventRecord(c->ev[3], st));
    // ---- stage 3: helpers + sweeps
    uint32_t *owner0, *owner1, *hcount, *hlo, *hhi, *ghost_head, *ghost_next; float *dist0, *dist1, *hc; unsigned char *R, *ghost_active, *ghost_done; int* ghost_vox;
    ENSURE(c->owner0, uint32_t, V, owner0); ENSURE(c->owner1, uint32_t, V, owner1); ENSURE(c->dist0, float, V, dist0); ENSURE(c->dist1, float, V, dist1);
    uint32_t* ownR_; ENSURE(c->ownR, uint32_t, V, ownR_);
    ENSURE(c->R, unsigned char, V, R); ENSURE(c->hc, float, (size_t)(S0 + 1) * 12, hc); ENSURE(c->hcount, uint32_t, S0 + 1, hcount);
    ENSURE(c->hlo, uint32_t, S0 + 1, hlo); ENSURE(c->hhi, uint32_t, S0 + 1, hhi); ENSURE(c->ghost_vox, int, S0 + 1, ghost_vox);
    ENSURE(c->ghost_active, unsigned char, S0 + 1, ghost_active); ENSURE(c->ghost_done, unsigned char, S0 + 1, ghost_done);
    ENSURE(c->ghost_head, uint32_t, V, ghost_head); ENSURE(c->ghost_next, uint32_t, S0 + 1, ghost_next);
    HIPCHECK(hipMemsetAsync(owner0, 0, (size_t)V * 4, st));
    HIPCHECK(hipMemsetAsync(ghost_head, 0, (size_t)V * 4, st));
    HIPCHECK(hipMemsetAsync(ghost_next, 0, (size_t)(S0 + 1) * 4, st));
    hipLaunchKernelGGL(k_fill_f32, dim3(grid_for(V, 256)), dim3(256), 0, st, dist0, V, F3DS_FLT_MAX);
    if (S0) hipLaunchKernelGGL(k_helper_own, dim3(grid_for(S0, 256)), dim3(256), 0, st, (const int*)seed_kept, S0, owner0);
    hipLaunchKernelGGL(k_helper_init, dim3(grid_for(S0 + 1, 256)), dim3(256), 0, st, (const int*)seed_kept, S0, (const uint32_t*)owner0, ghost_vox, ghost_active, ghost_done,
                       hlo, hhi, hcount, hc);
    HIPCHECK(hipMemsetAsync(R, 0, V, st));
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

// stage 3b: the label-propagation sweeps of every frame in `fr`, in lockstep, on one stream
int run_sweeps(std::vector<f3ds_ctx*>& fr, hipStream_t st) {
    if (fr.empty()) return F3DS_OK;
    f3ds_ctx* c0 = fr[0];
    const uint32_t sweeps = c0->res.sweeps;
    uint32_t maxV = 0, maxS = 0;
    for (f3ds_ctx* c : fr) { if (c->V > maxV) maxV = c->V; if (c->S0 > maxS) maxS = c->S0; }
    const uint32_t nf = (uint32_t)fr.size();
    SweepFrame* dargs;
    ENSURE(c0->sweep_args, SweepFrame, nf, dargs);
    std::vector<SweepFrame> args(nf);
    for (uint32_t t = 0; t < sweeps && maxS; ++t) {
        for (uint32_t i = 0; i < nf; ++i) {
            f3ds_ctx* c = fr[i];
            const f3ds_params& prm = c->prm;
            SweepFrame& a = args[i];
            a.sv = SweepView{(int)c->V, (const int*)c->nbrT.p, (const float*)c->vf.p, (const uint32_t*)c->owner0.p, (const float*)c->dist0.p, (const float*)c->hc.p,
                             (const uint32_t*)c->ghost_head.p, (const uint32_t*)c->ghost_next.p, (const uint32_t*)&c->d_dc->n_ghosts, prm.seed_res, prm.w_normal,
                             prm.w_color, prm.w_spatial};
            a.R = (unsigned char*)c->R.p; a.ownR = (uint32_t*)c->ownR.p; a.owner_out = (uint32_t*)c->owner1.p; a.dist_out = (float*)c->dist1.p;
            a.ghost_done = (unsigned char*)c->ghost_done.p; a.ghost_active = (unsigned char*)c->ghost_active.p; a.ghost_vox = (int*)c->ghost_vox.p;
            a.ghost_head = (uint32_t*)c->ghost_head.p; a.ghost_next = (uint32_t*)c->ghost_next.p;
            a.hlo = (uint32_t*)c->hlo.p; a.hhi = (uint32_t*)c->hhi.p; a.hcount = (uint32_t*)c->hcount.p; a.hc = (float*)c->hc.p; a.dc = c->d_dc; a.S0 = c->S0;
            if (t > 0 && a_sweep_needs_clear(t)) HIPCHECK(hipMemsetAsync(a.R, 0, c->V, st));
        }
        HIPCHECK(hipMemcpyAsync(dargs, args.data(), nf * sizeof(SweepFrame), hipMemcpyHostToDevice, st));
        const unsigned char tag = a_sweep_tag(t);
        hipLaunchKernelGGL(k_ghost_relink, dim3(1, nf), dim3(256), 0, st, (const SweepFrame*)dargs);
        hipLaunchKernelGGL(k_sweep_R, dim3(grid_for(maxV, 256), nf), dim3(256), 0, st, (const SweepFrame*)dargs, tag);
        hipLaunchKernelGGL(k_sweep_claim, dim3(grid_for(maxV, 256), nf), dim3(256), 0, st, (const SweepFrame*)dargs);
        hipLaunchKernelGGL(k_centroid, dim3(maxS, nf), dim3(64), 0, st, (const SweepFrame*)dargs);
        for (f3ds_ctx* c : fr) { std::swap(c->owner0, c->owner1); std::swap(c->dist0, c->dist1); }
    }
    for (f3ds_ctx* c : fr) HIPCHECK(hipEventRecord(c->ev[4], st));
    HIPCHECK(hipStreamSynchronize(st));
    return F3DS_OK;
}

// stage 4: supervoxel payload, adjacency, merge set-up (after the sweeps)
int segment_front_b(f3ds_ctx* c, const f3ds_params* prm) {
    HIPCHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const uint32_t V = c->V, S0 = c->S0;
    const uint32_t hcap = c->hmask + 1u;
    uint32_t *owner0 = (uint32_t*)c->owner0.p, *hcount = (uint32_t*)c->hcount.p, *hlo = (uint32_t*)c->hlo.p, *hhi = (uint32_t*)c->hhi.p;
    float *hc = (float*)c->hc.p, *vf = (float*)c->vf.p; int *ghost_vox = (int*)c->ghost_vox.p, *nbr = (int*)c->nbr.p;
    unsigned char* ghost_active = (unsigned char*)c->ghost_active.p;
    // ---- stage 4: supervoxel payload, adjacency
    uint32_t* loff; ENSURE(c->loff, uint32_t, S0 + 2, loff);
    { int rc = scan_u32(c, hcount, loff + 1, S0 + 1); if (rc) return rc; }     // loff[h+1] = inclusive => loff[h] = exclusive
    HIPCHECK(hipMemsetAsync(loff, 0, 4, st));
    // rows: one per leaf; ghosts add at most S0 to V
    float *rows, *racc0, *rrec0; int* row_voxel; uint32_t* rcnt0; unsigned char* ralive0;
    ENSURE(c->rows, float, ((size_t)V + S0 + 1) * 12, rows); ENSURE(c->row_voxel, int, (size_t)V + S0 + 1, row_voxel);
    ENSURE(c->racc0, float, (size_t)(S0 + 1) * 12, racc0); ENSURE(c->rcnt0, uint32_t, S0 + 1, rcnt0); ENSURE(c->rrec0, float, (size_t)(S0 + 1) * 16, rrec0);
    ENSURE(c->ralive0, unsigned char, S0 + 1, ralive0);
    HIPCHECK(hipMemsetAsync(ralive0, 0, S0 + 1, st));
    HIPCHECK(hipMemsetAsync(rcnt0, 0, (size_t)(S0 + 1) * 4, st));
    if (S0) hipLaunchKernelGGL(k_sv_fill, dim3(S0), dim3(64), 0, st, (const float*)vf, (const uint32_t*)owner0, S0, (const uint32_t*)hlo, (const uint32_t*)hhi, (const int*)ghost_vox,
                               (const unsigned char*)ghost_active, (const uint32_t*)hcount, (const uint32_t*)loff, (const float*)hc, rows, row_voxel, racc0, rcnt0, rrec0,
                               ralive0, c->d_dc);
    const uint32_t ecap = S0 * 32u + 1024u;
    const uint32_t ehcap = pow2_ge((size_t)ecap * 2);
    uint64_t *ehk, *ek0, *ek1; uint32_t *ev0, *ev1;
    ENSURE(c->hkeys, uint64_t, ehcap > hcap ? ehcap : hcap, ehk);      // the voxel hash is no longer needed
    ENSURE(c->ekeys0, uint64_t, ecap, ek0); ENSURE(c->ekeys1, uint64_t, ecap, ek1); ENSURE(c->evals0, uint32_t, ecap, ev0); ENSURE(c->evals1, uint32_t, ecap, ev1);
    HIPCHECK(hipMemsetAsync(ehk, 0xFF, (size_t)ehcap * 8, st));
    hipLaunchKernelGGL(k_edges, dim3(grid_for(V, 256)), dim3(256), 0, st, V, S0, (const int*)nbr, (const uint32_t*)owner0, ehk, ehcap - 1, ek0, ecap, c->d_dc);
    if (S0) hipLaunchKernelGGL(k_edges_ghost, dim3(grid_for(S0, 256)), dim3(256), 0, st, S0, (const int*)ghost_vox, (const unsigned char*)ghost_active, (const int*)nbr,
                               (const uint32_t*)owner0, ehk, ehcap - 1, ek0, ecap, c->d_dc);
    { int rc = sync_counters(c); if (rc) return rc; }
    HIPCHECK(hipGetLastError());
    if (c->h_dc->error) return c->h_dc->error;
    if (c->h_dc->r_overflow) return F3DS_ERR_UNSUPPORTED;
    const uint32_t E = c->h_dc->n_edges;
    c->E = E; c->res.n_edges = E; c->res.n_supervoxels = c->h_dc->n_alive;
    uint32_t *ea0, *eb0; ENSURE(c->ea0, uint32_t, E, ea0); ENSURE(c->eb0, uint32_t, E, eb0);
    if (E) {
        hipLaunchKernelGGL(k_iota, dim3(grid_for(E, 256)), dim3(256), 0, st, ev0, E);
        uint64_t* eks; uint32_t* evs;
        int rc = radix_sort(c, ek0, ev0, ek1, ev1, E, bits_for((uint64_t)(S0 + 1) * (S0 + 1)), &eks, &evs);
        if (rc) return rc;
        hipLaunchKernelGGL(k_edge_init, dim3(grid_for(E, 256)), dim3(256), 0, st, (const uint64_t*)eks, E, S0, ea0, eb0);
    }
    c->have_frame = true;
    int rc = run_cluster_front(c, prm);
    if (rc) { c->have_frame = false; return rc; }
    HIPCHECK(hipStreamSynchronize(st));      // the merge dispatch may run on another context's stream
    return 0;
}
void stage_times(f3ds_ctx* c, int first) {
    for (int i = first; i < 7; ++i) { float ms = 0; if (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) == hipSuccess) c->res.ms_stage[i] = ms; }
}
}  // namespace

extern "C" {

int f3ds_segment(f3ds_ctx* c, const void* points, size_t n_, int points_on_device, const f3ds_params* prm, uint32_t* point_labels,
                 int labels_on_device, f3ds_result* result) {
    const void* pp[1] = {points}; const size_t cnt[1] = {n_}; uint32_t* lp[1] = {point_labels};
    return f3ds_segment_batch(&c, 1, pp, cnt, points_on_device, prm, lp, labels_on_device, result);
}

// A batch of independent frames (BASELINE.json config 5: 8 frames per GPU).  Phases:
//   A1 per frame, own stream + host thread: voxelise, normals, seeds, helpers
//   A2 all frames in lockstep: the label-propagation sweeps as batched dispatches (grid.y = frame)
//   A3 per frame: supervoxel payload, adjacency, initial weights
//   B  all frames: the merge loops as ONE dispatch, one workgroup per frame (separate launches from
//      more than a handful of streams queue up behind each other on the compute pipes)
//   C  per frame: region ids, per-point labels, copy-out
int f3ds_segment_batch(f3ds_ctx** ctxs, int nctx, const void* const* points, const size_t* counts, int points_on_device, const f3ds_params* prm,
                       uint32_t* const* point_labels, int labels_on_device, f3ds_result* results) {
    if (!ctxs || nctx <= 0 || !points || !counts || !prm) return F3DS_ERR_ARG;
    for (int i = 0; i < nctx; ++i) if (!ctxs[i]) return F3DS_ERR_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<int> rcs((size_t)nctx, 0);
    auto parallel = [&](auto&& fn) {
        if (nctx == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (int i = 0; i < nctx; ++i) th.emplace_back([&, i]() { fn(i); });
        for (auto& t : th) t.join();
    };
    parallel([&](int i) { rcs[i] = segment_front_a(ctxs[i], points[i], counts[i], points_on_device, prm, point_labels ? point_labels[i] : nullptr, labels_on_device, nullptr); });
    std::vector<f3ds_ctx*> live;
    for (int i = 0; i < nctx; ++i) { if (rcs[i] < 0) return rcs[i]; if (rcs[i] == 0) live.push_back(ctxs[i]); }
    if (!live.empty()) {
        HIPCHECK(hipSetDevice(live[0]->device));
        int rc = run_sweeps(live, live[0]->stream);
        if (rc) return rc;
        parallel([&](int i) { if (rcs[i] == 0) rcs[i] = segment_front_b(ctxs[i], prm); });
        for (int i = 0; i < nctx; ++i) if (rcs[i] < 0) return rcs[i];
        HIPCHECK(hipSetDevice(live[0]->device));
        if ((rc = merge_launch(live.data(), (int)live.size(), live[0]->stream))) return rc;
        HIPCHECK(hipStreamSynchronize(live[0]->stream));
        parallel([&](int i) {
            if (rcs[i] != 0) return;
            (void)hipSetDevice(ctxs[i]->device);
            rcs[i] = run_cluster_tail(ctxs[i], prm, point_labels ? point_labels[i] : nullptr, labels_on_device);
            if (rcs[i]) ctxs[i]->have_frame = false; else stage_times(ctxs[i], 0);
        });
        for (int i = 0; i < nctx; ++i) if (rcs[i] < 0) return rcs[i];
    }
    const float ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < nctx; ++i) { ctxs[i]->res.ms_total = ms; if (results) results[i] = ctxs[i]->res; }
    return F3DS_OK;
}

int f3ds_recluster(f3ds_ctx* c, const f3ds_params* prm, uint32_t* point_labels, int labels_on_device, f3ds_result* result) {
    if (!c || !prm) return F3DS_ERR_ARG;
    if (!c->have_frame) return F3DS_ERR_LOGIC;
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHECK(hipSetDevice(c->device));
    c->h_dc->error = 0;
    HIPCHECK(hipMemsetAsync(&c->d_dc->error, 0, sizeof(int), c->stream));
    HIPCHECK(hipEventRecord(c->ev[4], c->stream));
    int rc = run_cluster(c, prm, point_labels, labels_on_device);
    if (rc) return rc;
    stage_times(c, 4);
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
}  // namespace

extern "C" int f3ds_get_voxel_cloud(f3ds_ctx* c, float* xyz, uint32_t* label, uint32_t* rgba, size_t cap, size_t* n_out) {
    if (!c) return F3DS_ERR_ARG;
    if (!c->have_frame) return F3DS_ERR_LOGIC;
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    const uint32_t S0 = c->S0;
    std::vector<unsigned char> ralive; std::vector<uint32_t> rhead, lnext, loff, llen, pool, rstart, rnleaf; std::vector<float> rows;
    int rc;
    if ((rc = fetch(c, c->ralive, S0 + 1, ralive)) || (rc = fetch(c, c->rhead, S0 + 1, rhead)) || (rc = fetch(c, c->lnext, S0 + 1, lnext)) ||
        (rc = fetch(c, c->loff, S0 + 2, loff)) || (rc = fetch(c, c->hcount, S0 + 1, llen)))
        return rc;
    if (c->merge_in_lds && ((rc = fetch(c, c->pool, c->pool.cap / 4, pool)) || (rc = fetch(c, c->rstart, S0 + 1, rstart)) || (rc = fetch(c, c->rnleaf, S0 + 1, rnleaf)))) return rc;
    if ((rc = fetch(c, c->rows, (size_t)loff[S0 + 1] * 12, rows))) return rc;
    size_t k = 0; uint32_t cur = 0;
    for (uint32_t h = 1; h <= S0; ++h) {
        if (!ralive[h]) continue;
        std::vector<uint32_t> leaves;      // the region's leaves in voxels_ concatenation order
        if (c->merge_in_lds) leaves.assign(pool.begin() + rstart[h], pool.begin() + rstart[h] + rnleaf[h]);
        else for (uint32_t leaf = rhead[h]; leaf; leaf = lnext[leaf]) leaves.push_back(leaf);
        for (uint32_t leaf : leaves)
            for (uint32_t j = 0; j < llen[leaf]; ++j) {
                if (k < cap) {
                    const float* r = &rows[(size_t)(loff[leaf] + j) * 12];
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
        case F3DS_DBG_GRID: { double g[5] = {c->h_grid->min[0], c->h_grid->min[1], c->h_grid->min[2], c->h_grid->res, (double)c->h_grid->depth}; put(g, sizeof g); break; }
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
                if (what == F3DS_DBG_SV_LABELS) put(&h, 4);
                else if (what == F3DS_DBG_SV_REGION) put(&u2[h], 4);
                else { float r[10]; for (int k = 0; k < 9; ++k) r[k] = f[(size_t)h * 12 + k]; r[9] = 0.0f; put(r, 40); }
            }
            break;
        case F3DS_DBG_EDGES:
            if ((rc = fetch(c, c->ea0, E, u)) || (rc = fetch(c, c->eb0, E, u2))) return rc;
            for (uint32_t e = 0; e < E; ++e) { put(&u[e], 4); put(&u2[e], 4); }
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
        case F3DS_DBG_MERGES: if ((rc = fetch(c, c->merges, (size_t)c->res.n_merges * 3, u))) return rc; put(u.data(), u.size() * 4); break;
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
