// f3ds_algo.h -- the data-parallel reformulation of the order-dependent parts of the path, as
// per-element functions shared by the HIP kernels (f3ds_hip.hip) and by the sequential CPU
// emulation used in the CPU test-suite (tests/emul/f3ds_emul.cpp).
//
// Three pieces of the reference are sequential as written and are restated here in a form whose
// result is identical but whose evaluation is independent per element:
//
//  (1) seed grid growth   (PCL OctreePointCloud::adoptBoundingBoxToPoint):
//      the cube only changes at a handful of "events"; a voxel's cell key is the key at its own
//      insertion epoch plus the integer shifts of the later events.
//  (2) label propagation  (PCL SupervoxelHelper::expand): helpers run in
//      label order inside a sweep and see each other's steals (Gauss-Seidel).  With
//      R(u) := "leaf u is still owned by its sweep-start owner when that owner's turn comes",
//         R(w) = not exists u in N(w): owner0(u) < owner0(w), d(owner0(u), w) < dist0(w), R(u)
//      (well-founded on the owner label), the state of voxel v after the sweep is obtained by
//      offering v, in ascending label order, to every helper that owns an R-true leaf in N(v).
//  (3) merge order        (std::multimap rebuild in Clustering::merge,
//      /root/reference/src/clustering.cpp:431-468): the new map is filled in old-map order, so
//      equal weights keep their previous relative order.  Map order is therefore the
//      lexicographic order of (w_t, w_{t-1}, ..., w_0, initial index); an edge only needs the
//      list of its own weight changes to be compared with any other edge.
#ifndef F3DS_ALGO_H_
#define F3DS_ALGO_H_

#include "f3ds_numerics.h"

namespace f3ds {

// ---------------------------------------------------------------------------------------------
// (1) seed grid
// ---------------------------------------------------------------------------------------------
#define F3DS_MAX_SEED_EVENTS 48
struct SeedEvent {
    int trigger;          // voxel index that caused the event
    int depth;            // tree depth after the event
    double min[3];        // cube minimum after the event
    unsigned off[3];      // key shift accumulated up to and including this event
};
struct SeedGrid {
    int n_events;
    int error;
    int defined;
    int depth;
    double res;
    double min[3], max[3];
    unsigned off[3];
    SeedEvent ev[F3DS_MAX_SEED_EVENTS];
};
F3DS_HD void a_seed_init(SeedGrid& g, float seed_res) {
    g.n_events = 0; g.error = 0; g.defined = 0; g.depth = 0; g.res = (double)seed_res;
    for (int a = 0; a < 3; ++a) { g.min[a] = g.max[a] = 0.0; g.off[a] = 0u; }
}
F3DS_HD bool a_seed_violates(const SeedGrid& g, const float p[3]) {
    if (!g.defined) return true;
    for (int a = 0; a < 3; ++a)
        if ((double)p[a] < g.min[a] || (double)p[a] >= g.max[a]) return true;
    return false;
}
// a box given as (lo, hi) corners lies inside the cube iff both corners do
F3DS_HD bool a_seed_box_violates(const SeedGrid& g, const float lo[3], const float hi[3]) {
    return a_seed_violates(g, lo) || a_seed_violates(g, hi);
}
F3DS_HD void a_seed_push(SeedGrid& g, int trigger) {
    if (g.n_events >= F3DS_MAX_SEED_EVENTS) { g.error = -4; return; }
    SeedEvent& e = g.ev[g.n_events++];
    e.trigger = trigger; e.depth = g.depth;
    for (int a = 0; a < 3; ++a) { e.min[a] = g.min[a]; e.off[a] = g.off[a]; }
}
// make point `trigger` fit: adoptBoundingBoxToPoint's while(true) loop
F3DS_HD void a_seed_grow(SeedGrid& g, int trigger, const float p[3]) {
    const double eps = (double)F3DS_FLT_EPS;
    while (!g.error) {
        bool lo[3], up[3];
        for (int a = 0; a < 3; ++a) { lo[a] = (double)p[a] < g.min[a]; up[a] = (double)p[a] >= g.max[a]; }
        if (!(lo[0] || lo[1] || lo[2] || up[0] || up[1] || up[2] || !g.defined)) break;
        if (g.defined) {
            double side = (double)(1u << g.depth) * g.res;
            unsigned shift = 1u << g.depth;
            for (int a = 0; a < 3; ++a)
                if (!up[a]) { g.min[a] -= side; g.off[a] += shift; }   // old root becomes the upper child
            g.depth++;
            if (g.depth > 21) { g.error = -4; return; }
            side = (double)(1u << g.depth) * g.res - eps;
            for (int a = 0; a < 3; ++a) g.max[a] = g.min[a] + side;
        } else {
            GridInfo t;
            t.res = g.res; t.error = 0;
            for (int a = 0; a < 3; ++a) { t.min[a] = (double)p[a] - g.res / 2; t.max[a] = (double)p[a] + g.res / 2; }
            n_key_bit_size(t);
            if (t.error) { g.error = t.error; return; }
            for (int a = 0; a < 3; ++a) { g.min[a] = t.min[a]; g.max[a] = t.max[a]; }
            g.depth = t.depth;
            g.defined = 1;
        }
        a_seed_push(g, trigger);
    }
}
// final cell key of voxel i (after all events)
F3DS_HD void a_seed_key(const SeedGrid& g, int i, const float p[3], unsigned key[3]) {
    int e = 0;
    for (int k = 0; k < g.n_events; ++k) if (g.ev[k].trigger <= i) e = k;
    const SeedEvent& ev = g.ev[e];
    for (int a = 0; a < 3; ++a)
        key[a] = (unsigned)(((double)p[a] - ev.min[a]) / g.res) + (g.off[a] - ev.off[a]);
}
F3DS_HD void a_seed_centre(const SeedGrid& g, const unsigned key[3], float c[3]) {
    for (int a = 0; a < 3; ++a) c[a] = (float)(((double)key[a] + 0.5f) * g.res + g.min[a]);
}
// flann::L2_Simple<float>
F3DS_HD float a_sqdist(const float* a, const float* b) {
    float r = 0.0f;
    float d = a[0] - b[0]; r += d * d;
    d = a[1] - b[1]; r += d * d;
    d = a[2] - b[2]; r += d * d;
    return r;
}
F3DS_HD float a_min_points(float seed_res, float voxel_res) {
    float search_radius = 0.5f * seed_res;
    return 0.05f * (search_radius) * (search_radius) * 3.1415926536f / (voxel_res * voxel_res);
}
F3DS_HD float a_radius_sq(float seed_res) {
    float search_radius = 0.5f * seed_res;
    return (float)((double)search_radius * (double)search_radius);
}

// ---------------------------------------------------------------------------------------------
// (2) label propagation sweep
// ---------------------------------------------------------------------------------------------
struct SweepView {
    int V;
    const int* nbrT;          // 27 x V (slot-major: lanes of consecutive voxels read consecutive words), -1 = none
    const float* vf;          // V x 12 voxel features: xyz rgb normal pad
    const uint32_t* owner;    // V, sweep-start owner label (0 = none)
    const float* dist;        // V, sweep-start VoxelData::distance_
    const float* hc;          // (S0+1) x 12 helper centroid features, row = label
    // "ghost" leaves: createSupervoxelHelpers puts the seed voxel into the helper's leaf set and
    // overwrites owner_; when two seeds resolve to the same voxel the earlier helper keeps a leaf
    // it does not own.  Such a leaf still expands and still counts in updateCentroid until the
    // helper steals it for real.  ghost_head[v] = first helper (label) with an active ghost on v,
    // ghost_next[label] chains further ones (0 ends the chain); *n_ghosts = active ghosts (the
    // chains are not even looked at once it is zero, which is the normal state after sweep 1).
    const uint32_t* ghost_head;   // V
    const uint32_t* ghost_next;   // S0+1
    const uint32_t* n_ghosts;
    float seed_res, w_normal, w_color, w_spatial;
    const int* nbr_rows = nullptr;      // V x 27, the same table row-major (optional): a voxel's 27 slots are two lines, not 27 -- for lanes that do NOT hold consecutive voxels
};
F3DS_HD int a_nbr(const SweepView& s, int v, int k) { return s.nbrT[(size_t)k * (size_t)s.V + (size_t)v]; }
// for the work-list kernels (chain walker, marks around a helper's leaves): their lanes hold scattered voxels, and what a gather costs is the lines it touches
F3DS_HD int a_nbr_w(const SweepView& s, int v, int k) { return s.nbr_rows ? s.nbr_rows[(size_t)v * 27u + (size_t)k] : a_nbr(s, v, k); }
// feature rows are 48 bytes, 16-byte aligned: three 128-bit loads instead of nine 32-bit ones
F3DS_HD void a_load_row(const float* p, float out[12]) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1], c = reinterpret_cast<const float4*>(p)[2];
    out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w; out[8] = c.x; out[9] = c.y; out[10] = c.z; out[11] = c.w;
#else
    for (int k = 0; k < 12; ++k) out[k] = p[k];
#endif
}
F3DS_HD float a_helper_dist(const SweepView& s, uint32_t g, int v) {
    return n_voxel_distance(s.hc + (size_t)g * 12, s.vf + (size_t)v * 12, s.seed_res, s.w_normal, s.w_color, s.w_spatial);
}
// the same with the voxel's row already loaded (the sweep kernels test one voxel against several helpers)
F3DS_HD float a_helper_dist_row(const SweepView& s, uint32_t g, const float vrow[12]) {
    float hrow[12];
    a_load_row(s.hc + (size_t)g * 12, hrow);
    return n_voxel_distance(hrow, vrow, s.seed_res, s.w_normal, s.w_color, s.w_spatial);
}
#define F3DS_R_STACK 24
#ifndef F3DS_R_ROUNDS
#define F3DS_R_ROUNDS 4        // rounds of the incremental R pass (dirty tiles): two do the work on most frames, but with three the last round still changed a word
#endif                         // in sweeps 8-10 of some of the 1M-point bench frames and sent them to the chain walker (0.26 ms per launch of 192 frames each); a change
                               // in the last one sends the sweep to the chain walker instead
#define F3DS_R_PASSES 1        // grid passes of the chain walker; what is still unsettled goes to the single-workgroup tail
#define F3DS_R_TRUE 1
#define F3DS_R_FALSE 2
#define F3DS_R_OPEN 3        // not settled yet, and no ghost leaf sits on the voxel or around it (written by the R pre-pass: the walker then skips its ghost-chain gathers there)
#define F3DS_OWNR_RTRUE 0x80000000u
// memo byte = (tag << 2) | value; tag = (sweep % 63) + 1 so that entries written in an earlier sweep
// read as "unknown" without clearing the array (it is zeroed before sweep 0, 63, 126, ...)
F3DS_HD unsigned char a_sweep_tag(unsigned sweep) { return (unsigned char)((sweep % 63u) + 1u); }
F3DS_HD bool a_sweep_needs_clear(unsigned sweep) { return sweep % 63u == 0u; }
// R(w) for an owned voxel w, memoised in `memo` (one byte per voxel).  Concurrent callers may race
// on memo entries: every writer stores the same value, and a stale "unknown" only costs a
// recomputation.  *overflow is set when the dependency chain is deeper than the explicit stack.
// a_eval_R_chain is the general form (explicit DFS stack); a_eval_R first scans w's neighbourhood
// with scalars only and falls into the chain walker just when some R(u) is really unknown, which
// keeps the stack arrays (scratch memory on the GPU) off the common path.
#if defined(__HIPCC__)
#define F3DS_NOINLINE __attribute__((noinline))
#else
#define F3DS_NOINLINE
#endif
F3DS_HD F3DS_NOINLINE bool a_eval_R_chain(const SweepView& s, int w0, unsigned char* memo, unsigned char tag, int* overflow) {
    const unsigned char T = (unsigned char)(tag << 2);
    const bool ghosts = *s.n_ghosts != 0u;
    int node[F3DS_R_STACK];
    int slot[F3DS_R_STACK];
    int sp = 0;
    node[0] = w0; slot[0] = 0;
    for (;;) {
        const int w = node[sp];
        const uint32_t h = s.owner[w];
        const float dw = s.dist[w];
        bool pushed = false, stolen = false;
        uint32_t g_cached = 0; bool cached_less = false;
        const bool ghosts_w = ghosts && memo[w] != (unsigned char)(T | F3DS_R_OPEN);      // (the pre-pass vouches for the voxels it marked open)
        for (int k = slot[sp]; k < 27; ++k) {
            int u = a_nbr_w(s, w, k);
            if (u < 0) continue;
            // a lower helper with a ghost leaf on u always reaches w at its turn
            if (ghosts_w) {
                for (uint32_t gg = s.ghost_head[u]; gg != 0u; gg = s.ghost_next[gg])
                    if (gg < h && a_helper_dist(s, gg, w) < dw) { stolen = true; break; }
                if (stolen) break;
            }
            uint32_t g = s.owner[u];
            if (g == 0u || g >= h) continue;
            if (g != g_cached) { g_cached = g; cached_less = a_helper_dist(s, g, w) < dw; }
            if (!cached_less) continue;
            // helper g steals w through u provided u is still g's at g's turn, i.e. R(u)
            const unsigned char mu = memo[u];
            if (mu == (T | F3DS_R_TRUE)) { stolen = true; break; }
            if (mu == (T | F3DS_R_FALSE)) continue;
            slot[sp] = k + 1;
            if (sp + 1 >= F3DS_R_STACK) { *overflow = 1; return true; }
            ++sp; node[sp] = u; slot[sp] = 0;
            pushed = true;
            break;
        }
        if (pushed) continue;
        bool r = !stolen;              // true: nobody steals node[sp] before its owner's turn
        for (;;) {
            memo[node[sp]] = (unsigned char)(T | (r ? F3DS_R_TRUE : F3DS_R_FALSE));
            if (sp == 0) return r;
            --sp;
            if (r) { r = false; continue; }   // child still owned -> parent stolen -> R(parent) = false
            break;                     // child was stolen first -> parent keeps scanning
        }
    }
}
F3DS_HD bool a_eval_R(const SweepView& s, int w, unsigned char* memo, unsigned char tag, int* overflow, bool ghost_near = true) {      // ghost_near = false: the caller knows that no ghost leaf sits on w or around it
    const unsigned char T = (unsigned char)(tag << 2);
    {
        const unsigned char m0 = memo[w];
        if ((m0 & 0xFC) == T && (m0 & 3) != F3DS_R_OPEN) return (m0 & 3) == F3DS_R_TRUE;
        if (m0 == (unsigned char)(T | F3DS_R_OPEN)) ghost_near = false;
    }
    if (ghost_near && *s.n_ghosts != 0u) return a_eval_R_chain(s, w, memo, tag, overflow);     // ghost leaves: general walker (it looks for them at every node it visits; the
                                                                                                  // scan below only leaves them out for w itself, where there are none)
    const uint32_t h = s.owner[w];
    const float dw = s.dist[w];
    // neighbours owned by a lower label (the only possible thieves before h's turn); every lane of a
    // wave first collects them, then the expensive distance is evaluated once per distinct owner
    // (loads are unconditional -- an absent neighbour reads w's own entry -- so that the 27 index loads and
    // then the 27 gathers are each in flight together instead of one round trip per neighbour)
    int nu[27]; uint32_t og[27];
    for (int k = 0; k < 27; ++k) nu[k] = a_nbr_w(s, w, k);
    for (int k = 0; k < 27; ++k) {
        const uint32_t g = s.owner[nu[k] >= 0 ? nu[k] : w];
        og[k] = (nu[k] >= 0 && g != 0u && g < h) ? g : 0u;
    }
    bool stolen = false, unknown = false;
    uint32_t last = 0;
    for (;;) {
        uint32_t g = 0xFFFFFFFFu;
        for (int k = 0; k < 27; ++k) if (og[k] > last && og[k] < g) g = og[k];
        if (g == 0xFFFFFFFFu) break;
        last = g;
        if (!(a_helper_dist(s, g, w) < dw)) continue;
        // g steals w through a leaf u of g that is still g's at g's turn, i.e. R(u)
        for (int k = 0; k < 27; ++k)
            if (og[k] == g) {
                const unsigned char mu = memo[nu[k]];
                if (mu == (T | F3DS_R_TRUE)) stolen = true;
                else if (mu != (T | F3DS_R_FALSE)) unknown = true;
            }
        if (stolen) break;
    }
    if (!stolen && unknown) return a_eval_R_chain(s, w, memo, tag, overflow);     // some R(u) still has to be derived
    memo[w] = (unsigned char)(T | (stolen ? F3DS_R_FALSE : F3DS_R_TRUE));
    return !stolen;
}
// Smallest of 27 label words that lies above `last`, as (label - last - 1); a result >= F3DS_NO_NEXT = none.  The subtraction does the filtering: a word
// at or below `last` -- including 0 = "no neighbour / no owner" -- wraps to a huge value and loses every minimum, so a candidate costs one subtract and
// half a three-input minimum instead of two compares, an and and a select (the sweeps' stencil kernels are bound by the VALU instructions they issue: one
// per four cycles per SIMD; the 27-wide search ran once per distinct candidate and was 3/4 of d_sweep_claim: DESIGN.md 4c).  `bias` = last + 1 for plain
// labels; for ownR words (label | R bit in bit 31) bias = 0x80000000 + last + 1: a word with its R bit set gives label - last - 1 as before, one without
// lands at or above 0x80000000 - last - 1.  Labels are below 2^30 (f3ds_segment refuses more seeds), so every rejected word reads >= 2^30 = F3DS_NO_NEXT.
#define F3DS_NO_NEXT 0x40000000u
F3DS_HD uint32_t a_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
F3DS_HD uint32_t a_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
F3DS_HD uint32_t a_next_label(const uint32_t x[27], uint32_t bias) {
    uint32_t m[9];
    for (int k = 0; k < 9; ++k) m[k] = a_umin(a_umin(x[3 * k] - bias, x[3 * k + 1] - bias), x[3 * k + 2] - bias);
    return a_umin(a_umin(a_umin(m[0], m[1]), m[2]), a_umin(a_umin(a_umin(m[3], m[4]), m[5]), a_umin(a_umin(m[6], m[7]), m[8])));
}
// (decision part, shared with the LDS-tiled kernels: x[k] = sweep-start owner of neighbour k, 0 for an absent one; h = w's own owner.  Labels are visited
// in ascending order and only those below h count: the search stops at the first one that is not)
// (w's distance and 48-byte feature row are fetched only once a lower label shows up among the words: 60-95 % of the voxels of a full sweep have none, and the rows
// of all voxels, 16 times per frame and kernel, were the largest single item of the sweeps' memory traffic)
F3DS_HD bool a_has_thief_words(const SweepView& s, int w, uint32_t h, const uint32_t x[27]) {
    uint32_t y = a_next_label(x, 1u);
    uint32_t g = y + 1u;
    if (y >= F3DS_NO_NEXT || g >= h) return false;
    const float dw = s.dist[w];
    float wrow[12];
    a_load_row(s.vf + (size_t)w * 12, wrow);
    for (;;) {
        if (a_helper_dist_row(s, g, wrow) < dw) return true;
        const uint32_t last = g;
        y = a_next_label(x, last + 1u);
        g = y + last + 1u;
        if (y >= F3DS_NO_NEXT || g >= h) return false;
    }
}
// ---- R through thief masks (round 6).  The pre-pass of a full sweep has the owners of w's 27 neighbours in registers and evaluates the distances d(g, w) of the
// lower labels g among them anyway: instead of "is there a thief at all" it now leaves, for every voxel it cannot settle, the MASK of the neighbour slots u whose
// owner g < owner0(w) has d(g, w) < dist0(w).  With it   R(w) = not exists k in mask(w): R(neighbour k of w)   -- the chain walker reads a word and a memo byte per
// candidate instead of gathering 27 owners, the helpers' centroid rows and w's feature row again (its gathers were the sweeps' third largest bill).
// x[k] = sweep-start owner of neighbour k (0 for an absent or unowned one), h = w's own owner.  Bit k of the result <=> neighbour k carries a thief.
F3DS_HD uint32_t a_thief_mask_words(const SweepView& s, int w, uint32_t h, const uint32_t x[27]) {
    uint32_t y = a_next_label(x, 1u);
    uint32_t g = y + 1u;
    if (y >= F3DS_NO_NEXT || g >= h) return 0u;
    const float dw = s.dist[w];
    float wrow[12];
    a_load_row(s.vf + (size_t)w * 12, wrow);
    uint32_t mask = 0u;
    for (;;) {
        if (a_helper_dist_row(s, g, wrow) < dw)
            for (int k = 0; k < 27; ++k) mask |= (x[k] == g ? 1u : 0u) << k;
        const uint32_t last = g;
        y = a_next_label(x, last + 1u);
        g = y + last + 1u;
        if (y >= F3DS_NO_NEXT || g >= h) return mask;
    }
}
// the same with the neighbours' owners gathered from global memory (tiles whose one-ring did not fit the LDS tables)
F3DS_HD uint32_t a_thief_mask(const SweepView& s, int w) {
    const uint32_t h = s.owner[w];
    int nu[27]; uint32_t x[27];
    for (int k = 0; k < 27; ++k) nu[k] = a_nbr(s, w, k);
    for (int k = 0; k < 27; ++k) {
        const uint32_t g = s.owner[nu[k] >= 0 ? nu[k] : w];           // unconditional loads, see a_eval_R
        x[k] = nu[k] >= 0 ? g : 0u;
    }
    return a_thief_mask_words(s, w, h, x);
}
// The ghost clause of R(w): a lower helper with a ghost leaf on a neighbour of w (or on w) reaches w at its turn whatever became of that leaf's owner -- no
// recursion.  The pre-pass only FLAGS the few voxels around a ghost leaf (bit 31 of their mask: F3DS_TMASK_GHOST; the flag alone puts a voxel on the walker's list)
// and the walker evaluates the clause when it first meets such a voxel: in the pre-pass kernel the loop below cost every voxel registers (17 spilled to scratch).
F3DS_HD bool a_ghost_steals(const SweepView& s, int w, uint32_t h) {
    const float dw = s.dist[w];
    for (int k = 0; k < 27; ++k) {
        const int u = a_nbr(s, w, k);
        if (u < 0) continue;
        for (uint32_t gg = s.ghost_head[u]; gg != 0u; gg = s.ghost_next[gg])
            if (gg < h && a_helper_dist(s, gg, w) < dw) return true;
    }
    return false;
}
// R(w) from the masks, memoised like a_eval_R (memo byte = (tag << 2) | value): first w's own candidates with scalars only, the explicit stack just when some
// R(u) is really unknown.  Every voxel that can show up unknown is owned and was not settled by the pre-pass, i.e. it has a mask of its own.
#define F3DS_TMASK_GHOST 0x80000000u
F3DS_HD bool a_eval_R_mask(const SweepView& s, int w0, unsigned char* memo, unsigned char tag, const uint32_t* tmask, int* overflow) {
    const unsigned char T = (unsigned char)(tag << 2);
    {
        const unsigned char m0 = memo[w0];
        if ((m0 & 0xFC) == T && (m0 & 3) != F3DS_R_OPEN) return (m0 & 3) == F3DS_R_TRUE;
    }
    uint32_t mask0 = tmask[w0];
    if (mask0 & F3DS_TMASK_GHOST) {
        if (a_ghost_steals(s, w0, s.owner[w0])) { memo[w0] = (unsigned char)(T | F3DS_R_FALSE); return false; }
        mask0 &= ~F3DS_TMASK_GHOST;
    }
    bool stolen = false, unknown = false;
    for (uint32_t m = mask0; m != 0u; m &= m - 1u) {
        const int u = a_nbr_w(s, w0, (int)__builtin_ctz(m));
        const unsigned char mu = memo[u];
        if (mu == (unsigned char)(T | F3DS_R_TRUE)) stolen = true;
        else if (mu != (unsigned char)(T | F3DS_R_FALSE)) unknown = true;
    }
    if (stolen || !unknown) { memo[w0] = (unsigned char)(T | (stolen ? F3DS_R_FALSE : F3DS_R_TRUE)); return !stolen; }
    int node[F3DS_R_STACK];
    uint32_t rem[F3DS_R_STACK];
    int sp = 0;
    node[0] = w0; rem[0] = mask0;
    for (;;) {
        const int w = node[sp];
        bool pushed = false, st = false;
        while (rem[sp] != 0u) {
            const int k = (int)__builtin_ctz(rem[sp]);
            const int u = a_nbr_w(s, w, k);
            const unsigned char mu = memo[u];
            if (mu == (unsigned char)(T | F3DS_R_TRUE)) { st = true; break; }
            rem[sp] &= rem[sp] - 1u;
            if (mu == (unsigned char)(T | F3DS_R_FALSE)) continue;
            if (sp + 1 >= F3DS_R_STACK) { *overflow = 1; return true; }
            // R(u) still has to be derived: u's bit is gone from w's remaining mask -- when u comes back R-true w is stolen, when it comes back R-false w goes on
            uint32_t mu_ = tmask[u];
            if (mu_ & F3DS_TMASK_GHOST) {
                if (a_ghost_steals(s, u, s.owner[u])) { memo[u] = (unsigned char)(T | F3DS_R_FALSE); continue; }      // (R(u) is false: w goes on)
                mu_ &= ~F3DS_TMASK_GHOST;
            }
            ++sp; node[sp] = u; rem[sp] = mu_;
            pushed = true;
            break;
        }
        if (pushed) continue;
        bool r = !st;
        for (;;) {
            memo[node[sp]] = (unsigned char)(T | (r ? F3DS_R_TRUE : F3DS_R_FALSE));
            if (sp == 0) return r;
            --sp;
            if (r) { r = false; continue; }   // child still owned -> parent stolen -> R(parent) = false
            break;                     // child was stolen first -> parent keeps scanning
        }
    }
}
// One application of the defining equation of R to voxel w, reading the neighbours' R from bit 31 of
// ownR instead of deriving it: the incremental sweeps (f3ds_kernels.inc, "dirty tiles") iterate this
// to the fixed point, which is unique because R(w) only depends on R of voxels with a lower owner.
// Not usable while ghost leaves are active (those sweeps run the chain walker).
F3DS_HD bool a_eval_R_step(const SweepView& s, const uint32_t* ownR, int w) {
    const uint32_t h = s.owner[w];
    if (h == 0u) return false;
    int nu[27]; uint32_t x[27];
    for (int k = 0; k < 27; ++k) nu[k] = a_nbr(s, w, k);
    for (int k = 0; k < 27; ++k) {
        const int u = nu[k] >= 0 ? nu[k] : w;        // unconditional loads, see a_eval_R
        const uint32_t g = s.owner[u], r = ownR[u];
        x[k] = (nu[k] >= 0 && (r & F3DS_OWNR_RTRUE)) ? g : 0u;       // helper that still holds u at its turn (as far as ownR knows)
    }
    return !a_has_thief_words(s, w, h, x);
}
// Can any lower helper steal w at all, whatever the R of its leaves?  False means R(w) holds without looking at
// any other voxel; the full sweeps settle most voxels this way and run the chain walker on the rest only.
// (Same tests as a_eval_R_step with every neighbour's R taken as true; not for sweeps with ghost leaves.)
F3DS_HD bool a_has_thief(const SweepView& s, int w) {
    const uint32_t h = s.owner[w];
    int nu[27]; uint32_t x[27];
    for (int k = 0; k < 27; ++k) nu[k] = a_nbr(s, w, k);
    for (int k = 0; k < 27; ++k) {
        const uint32_t g = s.owner[nu[k] >= 0 ? nu[k] : w];           // unconditional loads, see a_eval_R
        x[k] = nu[k] >= 0 ? g : 0u;
    }
    return a_has_thief_words(s, w, h, x);
}
// state of voxel v after the sweep.  ownR[u] = sweep-start owner of u with bit 31 set when R(u) holds
// (written by the R pass for every voxel), so a neighbour costs one gather.  ghost_done[g] is set
// when helper g turns its ghost leaf on v into a real one (only the thread of v writes it).
// (decision part without ghost leaves, shared with the LDS-tiled kernels: x[k] = the ownR word of neighbour k -- its sweep-start owner, bit 31 set when
// that owner still holds it at its turn --, 0 for an absent neighbour: the helpers that offer v are the labels of the words with bit 31 set)
F3DS_HD void a_claim_words(const SweepView& s, int v, const uint32_t x[27], uint32_t* owner_out, float* dist_out) {
    uint32_t o = s.owner[v];
    float d = s.dist[v];
#if defined(F3DS_EXP_CLAIM_NOSEL)      // timing experiment: no candidate search at all (wrong results)
    { uint32_t z = 0; for (int k = 0; k < 27; ++k) z |= x[k]; *owner_out = z == 0x7FFFFFFFu ? 1u : o; *dist_out = d; return; }
#endif
    float vrow[12];
    a_load_row(s.vf + (size_t)v * 12, vrow);      // (fetched up front also for interior voxels, which never use it: fetched at the first foreign offer the kernel is 6 % slower for 6 MB per frame less)
    uint32_t last = 0;
    for (;;) {
        const uint32_t y = a_next_label(x, F3DS_OWNR_RTRUE + last + 1u);       // smallest offering label above `last`
        if (y >= F3DS_NO_NEXT) break;
        const uint32_t g = y + last + 1u;
        last = g;
        if (g == o) continue;          // neighbor_voxel.owner_ == this
#if defined(F3DS_EXP_CLAIM_NODIST)     // timing experiment: candidates searched, no distance evaluated (wrong results)
        float dg = (float)g + vrow[0];
#else
        float dg = a_helper_dist_row(s, g, vrow);
#endif
        if (dg < d) { d = dg; o = g; }
    }
    *owner_out = o; *dist_out = d;
}
F3DS_HD void a_claim(const SweepView& s, const uint32_t* ownR, int v, uint32_t* owner_out, float* dist_out, unsigned char* ghost_done) {
    const bool ghosts = *s.n_ghosts != 0u;
    int nu[27]; uint32_t cand[27];
    for (int k = 0; k < 27; ++k) nu[k] = a_nbr(s, v, k);
    for (int k = 0; k < 27; ++k) {
        const uint32_t x = ownR[nu[k] >= 0 ? nu[k] : v];               // unconditional loads, see a_eval_R
        cand[k] = nu[k] >= 0 ? x : 0u;
    }
    if (!ghosts) { a_claim_words(s, v, cand, owner_out, dist_out); return; }
    for (int k = 0; k < 27; ++k) cand[k] = (cand[k] & F3DS_OWNR_RTRUE) ? (cand[k] & 0x7fffffffu) : 0u;      // helper that offers v through leaf u (0 = none)
    uint32_t o = s.owner[v];
    float d = s.dist[v];
    float vrow[12];
    a_load_row(s.vf + (size_t)v * 12, vrow);
    uint32_t last = 0;
    for (;;) {
        uint32_t g = 0xFFFFFFFFu;     // smallest candidate label above `last`
        for (int k = 0; k < 27; ++k) {
            const uint32_t gu = cand[k];
            if (gu > last && gu < g) g = gu;
        }
        for (int k = 0; k < 27; ++k) {
            const int u = a_nbr(s, v, k);
            if (u < 0) continue;
            for (uint32_t gg = s.ghost_head[u]; gg != 0u; gg = s.ghost_next[gg])
                if (gg > last && gg < g) g = gg;
        }
        if (g == 0xFFFFFFFFu) break;
        last = g;
        if (g == o) continue;          // neighbor_voxel.owner_ == this
        float dg = a_helper_dist_row(s, g, vrow);
        if (dg < d) {
            d = dg; o = g;
            for (uint32_t gg = s.ghost_head[v]; gg != 0u; gg = s.ghost_next[gg])
                if (gg == g) ghost_done[g] = 1;
        }
    }
    *owner_out = o; *dist_out = d;
}
// SupervoxelHelper::updateCentroid from the ordered sums {xyz, rgb, normal xyz}
F3DS_HD void a_centroid_finish(const float sum[9], unsigned count, float row[12]) {
    float nx = sum[6], ny = sum[7], nz = sum[8];
    float z = (nx * nx + ny * ny) + (nz * nz + 0.0f);
    if (z > 0.0f) { float q = n_sqrtf(z); nx /= q; ny /= q; nz /= q; }
    float c = (float)count;
    row[0] = sum[0] / c; row[1] = sum[1] / c; row[2] = sum[2] / c;
    row[3] = sum[3] / c; row[4] = sum[4] / c; row[5] = sum[5] / c;
    row[6] = nx; row[7] = ny; row[8] = nz;
    row[9] = row[10] = row[11] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// (3) merge stage
// ---------------------------------------------------------------------------------------------
// payload row of one voxel inside a supervoxel: xx xy xz yy yz zz x y z r g b
F3DS_HD void a_payload_row(const float* vf_row, float out[12]) {
    float x = vf_row[0], y = vf_row[1], z = vf_row[2];
    out[0] = x * x; out[1] = x * y; out[2] = x * z; out[3] = y * y; out[4] = y * z; out[5] = z * z;
    out[6] = x; out[7] = y; out[8] = z;
    // VoxelData::getPoint truncates the mean colour to 8 bits; mean_color reads it back as float
    out[9] = (float)((uint32_t)vf_row[3] & 255u);
    out[10] = (float)((uint32_t)vf_row[4] & 255u);
    out[11] = (float)((uint32_t)vf_row[5] & 255u);
}
// one step of the ordered fold: acc[0..8] += row, acc[9..11] running mean with count = index+1
F3DS_HD void a_fold_row(float acc[12], const float row[12], unsigned index_plus_1) {
    for (int k = 0; k < 9; ++k) acc[k] += row[k];
    float count = (float)index_plus_1;
    float inv = 1 / count;
    for (int k = 9; k < 12; ++k) acc[k] = acc[k] + inv * (row[k] - acc[k]);
}
// region record (16 floats: centroid, normal, mean rgb, Lab) of a merged region from its sums
F3DS_HD void a_region_from_acc(const float acc[12], unsigned count, float rec[16]) {
    float c = (float)count;
    rec[0] = acc[6] / c; rec[1] = acc[7] / c; rec[2] = acc[8] / c;
    float n4[4];
    n_plane_normal(acc, count, rec, n4);
    rec[3] = n4[0]; rec[4] = n4[1]; rec[5] = n4[2];
    rec[6] = acc[9]; rec[7] = acc[10]; rec[8] = acc[11];
    n_rgb2lab(rec + 6, rec + 9);
    rec[12] = rec[13] = rec[14] = rec[15] = 0.0f;
}

struct MergeParams {
    int color_metric, geom_metric, merging;
    float lambda;
    int bins;
    const float* cdf_c;       // bins entries (EQUALIZATION)
    const float* cdf_g;
};
// t_c / t_g (src/clustering.cpp:324-376); *err set when the reference's map::at would throw
F3DS_HD float a_tc(const MergeParams& p, float dc, int* err) {
    if (p.merging != 2) return p.lambda * dc;
    short bin = (short)__builtin_floorf(dc * (float)(short)p.bins);
    if (bin == (short)p.bins) bin--;
    if (bin < 0 || bin >= (short)p.bins) { *err = -9; return 0.0f; }
    return p.cdf_c[bin] / 2;
}
F3DS_HD float a_tg(const MergeParams& p, float dg, int* err) {
    if (p.merging != 2) return (1 - p.lambda) * dg;
    short bin = (short)__builtin_floorf(dg * (float)(short)p.bins);
    if (bin < 0 || bin >= (short)p.bins) { *err = -9; return 0.0f; }
    return p.cdf_g[bin] / 2;
}
template <class K = m_lit> F3DS_HD float a_edge_weight(const MergeParams& p, const float* rec_first, const float* rec_second, int* err, K mc = K()) {
    float dc, dg;
    n_delta_c_g(rec_first, rec_second, p.color_metric, p.geom_metric, &dc, &dg, mc);
    return a_tc(p, dc, err) + a_tg(p, dg, err);
}

// weight-change history of the edges: event h = (epoch, key, previous event of the same edge)
struct EdgeHist {
    const uint32_t* ev_epoch;
    const uint32_t* ev_key;
    const int* ev_prev;
};
// true when edge e precedes edge f in the reference's weight_map (e != f)
F3DS_HD bool a_edge_before(const EdgeHist& H, uint32_t e, uint32_t ke, int he, uint32_t f, uint32_t kf, int hf) {
    if (ke != kf) return ke < kf;
    for (;;) {
        uint32_t te = H.ev_epoch[he], tf = H.ev_epoch[hf];
        uint32_t tau = te > tf ? te : tf;
        if (tau == 0u) return e < f;       // initial insertion order = sorted (a,b) order
        if (te == tau) he = H.ev_prev[he];
        if (tf == tau) hf = H.ev_prev[hf];
        ke = H.ev_key[he]; kf = H.ev_key[hf];
        if (ke != kf) return ke < kf;
    }
}

}  // namespace f3ds
#endif  // F3DS_ALGO_H_
