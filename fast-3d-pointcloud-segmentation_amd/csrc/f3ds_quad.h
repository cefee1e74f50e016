// f3ds_quad.h -- device-only pieces of the merge loop's re-weighting that the micro-benchmarks time on their own (tools/ubench/ubench_math.hip):
// CIEDE2000 on the four lanes of a quad, and the edge weight built on it.  Included by f3ds_kernels.inc inside its anonymous namespace.
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
__device__ inline float n_ciede00_quad(const float lab1[3], const float lab2[3], int q) {
    const double PI = 3.14159265358979323846;
    const double P25_7 = 6103515625.0;
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
    double hpm = 0.0;
    if ((m_abs(apm) + (double)m_absf(bm)) != 0.0) {
        hpm = m_atan2((double)bm, apm);
        if (hpm < 0) hpm += 2.0 * PI;
    }
    const double Cp1 = quad_bcast<0>(Cpm), Cp2 = quad_bcast<1>(Cpm);
    const double hp1 = quad_bcast<0>(hpm), hp2 = quad_bcast<1>(hpm);
    const double Cp_prod = Cp2 * Cp1;
    const double dL = (double)(L2 - L1);
    const double dC = Cp2 - Cp1;
    double dhp = hp2 - hp1;
    if (dhp > PI) dhp -= 2.0 * PI;
    else if (dhp < -PI) dhp += 2.0 * PI;
    if (Cp_prod == 0.0) dhp = 0.0;
    const double Lp = (double)(L2 + L1) / 2.0;
    const double Cp = (Cp1 + Cp2) / 2.0;
    double hp = (hp1 + hp2) / 2.0;
    if (m_abs(hp1 - hp2) > PI) hp -= PI;
    if (hp < 0) hp += 2.0 * PI;
    if (Cp_prod == 0.0) hp = hp1 + hp2;
    const double Lpm502 = (Lp - 50.0) * (Lp - 50.0);
    // the four cosines of T, one per lane
    const double carg = q == 0 ? hp - PI / 6.0 : (q == 1 ? 2.0 * hp : (q == 2 ? 3.0 * hp + PI / 30.0 : 4.0 * hp - 63.0 * PI / 180.0));
    const double cq = m_cos(carg);
    const double T = 1.0 - 0.17 * quad_bcast<0>(cq) + 0.24 * quad_bcast<1>(cq) + 0.32 * quad_bcast<2>(cq) - 0.20 * quad_bcast<3>(cq);
    const double e = (180.0 / PI * hp - 275.0) / 25.0;
    const double dtheta = (30.0 * PI / 180.0) * m_exp(-(e * e));
    const double Cp7 = n_pow7(Cp);
    // sqrt(Cp_prod) | sqrt(Cp7 / (Cp7 + 25^7)) | sqrt(20 + (Lp - 50)^2) in lanes 0 | 1 | 2
    const double sarg = q == 0 ? Cp_prod : (q == 1 ? Cp7 / (Cp7 + P25_7) : 20.0 + Lpm502);
    const double sq = n_sqrt(sarg);
    // sin(dh'/2) | sin(2 dtheta) in lanes 0 | 1
    const double sn = m_sin(q == 0 ? dhp / 2.0 : 2.0 * dtheta);
    const double dH = 2.0 * quad_bcast<0>(sq) * quad_bcast<0>(sn);
    const double Rc = 2.0 * quad_bcast<1>(sq);
    const double kLSL = 1.0 * (1.0 + 0.015 * Lpm502 / quad_bcast<2>(sq));
    const double kLSC = 1.0 * (1.0 + 0.045 * Cp);
    const double kHSH = 1.0 * (1.0 + 0.015 * Cp * T);
    const double RT = -quad_bcast<1>(sn) * Rc;
    const double tL = dL / kLSL, tC = dC / kLSC, tH = dH / kHSH;
    return (float)n_sqrt(tL * tL + tC * tC + tH * tH + RT * tC * tH);
}
// a_edge_weight with the quad version of the colour distance (LAB_CIEDE00 only)
__device__ inline float edge_weight_quad(const MergeParams& p, const float* r1, const float* r2, int q, int* err) {
    const float dc = n_ciede00_quad(r1 + 9, r2 + 9, q) / F3DS_LAB_RANGE;
    float dg = n_normals_diff(r1 + 3, r1, r2 + 3, r2);
    if (p.geom_metric == 1 && n_is_convex(r1 + 3, r1, r2 + 3, r2)) dg *= 0.5;
    return a_tc(p, dc, err) + a_tg(p, dg, err);
}
#endif  // F3DS_QUAD_H_
