// f3ds_quad.h -- device-only pieces of the merge loop's re-weighting that the micro-benchmarks time on their own (tools/ubench/ubench_math.hip):
// CIEDE2000 on the four lanes of a quad, the edge weight built on it, rgb -> Lab on three lanes.  Included by f3ds_kernels.inc inside its anonymous namespace.
#ifndef F3DS_QUAD_H_
#define F3DS_QUAD_H_
// ------------------------------------------------------------------------------------------------
// CIEDE2000 spread over the four lanes of a quad (the merge loop re-weights a few dozen edges per merge
// and is otherwise idle: one lane per edge leaves the ~20 transcendental sequences of n_ciede00 in a row).
// Every sub-expression is the one of n_ciede00 (f3ds_numerics.h), on the same operands, in the same
// order -- only WHICH lane evaluates it changes: the two atan2 run side by side in lanes 0/1, the four
// cosines of T in lanes 0..3, sin(dh'/2) beside sin(2 dtheta), the three later square roots together.
// All four lanes must be active and hold the same lab1 / lab2; all return the same result.
// ------------------------------------------------------------------------------------------------
// the f64 constants of f3ds_math.h / n_ciede00 read from a copy of M_TABLE in LDS (see f3ds_math.h "where the f64 constants come from")
struct m_lds {
    __attribute__((address_space(3))) const double* t;
    __device__ __forceinline__ double operator()(int i) const { return t[i]; }
};
// A provider whose table address the compiler cannot see through: the loads behind it stay where the arithmetic is.  (The table never changes, so with a plain pointer
// every constant is loop-invariant and is hoisted out of the merge loop: ~100 VGPRs held for good, the loop's own state spilled to scratch.)
__device__ __forceinline__ m_lds m_lds_here(const double* tab) {
    __attribute__((address_space(3))) const double* p = (__attribute__((address_space(3))) const double*)tab;
    asm volatile("" : "+v"(p));
    return m_lds{p};
}
template <int K>
__device__ inline double quad_bcast(double x) {
    constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);          // quad_perm [K,K,K,K]
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xFFFFFFFFll), ctrl, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), ctrl, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
template <int K>
__device__ inline float quad_bcastf(float x) {
    constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true));
}
template <class MC = m_lit>
__device__ inline float n_ciede00_quad(const float lab1[3], const float lab2[3], int q, MC mc = MC()) {
    const double PI = mc(MC_CIE_PI), TWO_PI = mc(MC_CIE_2PI);
    const double P25_7 = mc(MC_CIE_25_7);
    const float L1 = lab1[0], a1 = lab1[1], b1 = lab1[2];
    const float L2 = lab2[0], a2 = lab2[1], b2 = lab2[2];
    // lanes 0 / 1 carry colour 1 / colour 2 through the per-colour chain (lanes 2, 3 repeat colour 2)
    const float am = q == 0 ? a1 : a2, bm = q == 0 ? b1 : b2;
    const float cabf = n_sqrtf(am * am + bm * bm);
    const double Cab1 = (double)quad_bcastf<0>(cabf), Cab2 = (double)quad_bcastf<1>(cabf);
    const double Cab = (Cab1 + Cab2) / 2.0;
    const double Cab7 = n_pow7(Cab);
    const double G = 0.5 * (1.0 - n_sqrt(Cab7 / (Cab7 + P25_7)));
    const double apm = (1.0 + G) * (double)am;
    const double Cpm = n_sqrt(apm * apm + (double)(bm * bm));
    double hpm = m_atan2((double)bm, apm, mc);      // (evaluated for every lane, selected: no branch around it)
    hpm = hpm < 0 ? hpm + TWO_PI : hpm;
    hpm = (m_abs(apm) + (double)m_absf(bm)) != 0.0 ? hpm : 0.0;
    const double Cp1 = quad_bcast<0>(Cpm), Cp2 = quad_bcast<1>(Cpm);
    const double hp1 = quad_bcast<0>(hpm), hp2 = quad_bcast<1>(hpm);
    const double Cp_prod = Cp2 * Cp1;
    const double dL = (double)(L2 - L1);
    const double dC = Cp2 - Cp1;
    double dhp = hp2 - hp1;
    if (dhp > PI) dhp -= TWO_PI;
    else if (dhp < -PI) dhp += TWO_PI;
    if (Cp_prod == 0.0) dhp = 0.0;
    const double Lp = (double)(L2 + L1) / 2.0;
    const double Cp = (Cp1 + Cp2) / 2.0;
    double hp = (hp1 + hp2) / 2.0;
    if (m_abs(hp1 - hp2) > PI) hp -= PI;
    if (hp < 0) hp += TWO_PI;
    if (Cp_prod == 0.0) hp = hp1 + hp2;
    const double Lpm502 = (Lp - mc(MC_CIE_50)) * (Lp - mc(MC_CIE_50));
    // the four cosines of T, one per lane
    const double carg = q == 0 ? hp - mc(MC_CIE_PI_6) : (q == 1 ? 2.0 * hp : (q == 2 ? mc(MC_CIE_3) * hp + mc(MC_CIE_PI_30) : 4.0 * hp - mc(MC_CIE_63PI_180)));
    const double cq = m_cos(carg, mc);
    const double T = 1.0 - mc(MC_CIE_017) * quad_bcast<0>(cq) + mc(MC_CIE_024) * quad_bcast<1>(cq) + mc(MC_CIE_032) * quad_bcast<2>(cq) - mc(MC_CIE_020) * quad_bcast<3>(cq);
    const double e = (mc(MC_CIE_180_PI) * hp - mc(MC_CIE_275)) / mc(MC_CIE_25);
    const double dtheta = mc(MC_CIE_30PI_180) * m_exp(-(e * e), mc);
    const double Cp7 = n_pow7(Cp);
    // sqrt(Cp_prod) | sqrt(Cp7 / (Cp7 + 25^7)) | sqrt(20 + (Lp - 50)^2) in lanes 0 | 1 | 2
    const double sarg = q == 0 ? Cp_prod : (q == 1 ? Cp7 / (Cp7 + P25_7) : mc(MC_CIE_20) + Lpm502);
    const double sq = n_sqrt(sarg);
    // sin(dh'/2) | sin(2 dtheta) in lanes 0 | 1
    const double sn = m_sin(q == 0 ? dhp / 2.0 : 2.0 * dtheta, mc);
    const double dH = 2.0 * quad_bcast<0>(sq) * quad_bcast<0>(sn);
    const double Rc = 2.0 * quad_bcast<1>(sq);
    const double kLSL = 1.0 * (1.0 + mc(MC_CIE_0015) * Lpm502 / quad_bcast<2>(sq));
    const double kLSC = 1.0 * (1.0 + mc(MC_CIE_0045) * Cp);
    const double kHSH = 1.0 * (1.0 + mc(MC_CIE_0015) * Cp * T);
    const double RT = -quad_bcast<1>(sn) * Rc;
    const double tL = dL / kLSL, tC = dC / kLSC, tH = dH / kHSH;
    return (float)n_sqrt(tL * tL + tC * tC + tH * tH + RT * tC * tH);
}
// a_edge_weight with the quad version of the colour distance (LAB_CIEDE00 only)
template <class MC = m_lit>
__device__ inline float edge_weight_quad(const MergeParams& p, const float* r1, const float* r2, int q, int* err, MC mc = MC()) {
    const float dc = n_ciede00_quad(r1 + 9, r2 + 9, q, mc) / F3DS_LAB_RANGE;
    float dg = n_normals_diff(r1 + 3, r1, r2 + 3, r2);
    if (p.geom_metric == 1 && n_is_convex(r1 + 3, r1, r2 + 3, r2)) dg *= 0.5;
    return a_tc(p, dc, err) + a_tg(p, dg, err);
}
// n_rgb2lab (f3ds_numerics.h) for a wave whose lane k < 3 holds the mean of colour channel k in `mine` (the other lanes repeat channel 2): the three gamma
// curves and the three cube roots are evaluated side by side, everything else as there.  Every lane returns L, a, b.
// Gamma curve and cube root are evaluated for every lane and the linear pieces selected afterwards (same values: no branch, no divergence between the three lanes).
// (base: the first of the three lanes -- 0 for a wave, 16 r for row r of a wave that holds four colours, one per row; `lane` counts from it)
template <class MC = m_lit>
__device__ inline void lab_three_lanes(float mine, int lane, float lab[3], MC mc = MC(), int base = 0) {
    const float v = mine / 255;
    const float gam = (float)m_pow_pos((double)((v + 0.055f) / 1.055f), mc(MC_GAMMA_EXP), mc);
    const float cl = v <= 0.04045f ? v / 12.92f : gam;
    const float c0 = __shfl(cl, base, 64), c1 = __shfl(cl, base + 1, 64), c2 = __shfl(cl, base + 2, 64);
    const float X = (c0 * 0.412453f + c1 * 0.357580f + c2 * 0.180423f) / 0.950456f;
    const float Y = (c0 * 0.212671f + c1 * 0.715160f + c2 * 0.072169f);
    const float Z = (c0 * 0.019334f + c1 * 0.119193f + c2 * 0.950227f) / 1.088754f;
    const float t = lane == 0 ? X : (lane == 1 ? Y : Z);
    const float cb = (float)m_cbrt_pos((double)t, mc);
    const float fl = t > 0.008856f ? cb : 7.787f * t + 16.0f / 116.0f;          // n_lab_f
    const float fx = __shfl(fl, base, 64), fy = __shfl(fl, base + 1, 64), fz = __shfl(fl, base + 2, 64);
    lab[0] = Y > 0.008856f ? 116.0f * fy - 16.0f : 903.3f * Y;
    lab[1] = 500.0f * (fx - fy);
    lab[2] = 200.0f * (fy - fz);
}
#endif  // F3DS_QUAD_H_
