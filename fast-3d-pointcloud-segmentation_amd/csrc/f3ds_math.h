// f3ds_math.h -- transcendental functions built from IEEE-754 basic operations only.
//
// Why this exists: the segmentation path takes a logarithm before voxel keys are
// generated (single-camera transform) and uses atan2/cos/sin/exp inside the merge
// distance.  libm (host) and ocml (device) do not return the same last bit for those,
// and one flipped bit can move a point to the neighbouring voxel or reorder two merges.
// Every function below uses only + - * / sqrt fma, integer bit moves and comparisons, so
// the same source gives the same bits from g++ (x86-64) and from hipcc (gfx950),
// provided both are compiled with -ffp-contract=off (the build scripts do that; the host
// builds add -mfma so that __builtin_fma is the instruction, glibc's fma() gives the same bits).
//
// Replaces, at the reference's call sites:
//   std::log(float)              PCL SupervoxelClustering::transformFunction
//   std::atan2/cos/sin (float)   pcl::computeRoots                          
//   std::atan2/cos/sin/exp/pow   ColorUtilities::lab_ciede00  /root/reference/src/color_utilities.cpp:200-291
//   cv::cvtColor gamma + cbrt    ColorUtilities::rgb2lab      /root/reference/src/color_utilities.cpp:151-160
//
// Accuracy (checked in tests/test_math.py against libm): <= 2 ulp in double over the
// argument ranges the path uses; the float wrappers round a double result and differ
// from glibc's float functions in well under 1e-6 of random arguments.
#ifndef F3DS_MATH_H_
#define F3DS_MATH_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define F3DS_HD __host__ __device__ inline
#else
#define F3DS_HD inline
#endif

namespace f3ds {

F3DS_HD uint64_t m_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
F3DS_HD double m_from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
F3DS_HD uint32_t m_bitsf(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
F3DS_HD float m_from_bitsf(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

F3DS_HD double m_abs(double x) { return m_from_bits(m_bits(x) & 0x7fffffffffffffffULL); }
F3DS_HD float m_absf(float x) { return m_from_bitsf(m_bitsf(x) & 0x7fffffffu); }
F3DS_HD bool m_isnan(double x) { return x != x; }
F3DS_HD bool m_isinf(double x) { return (m_bits(x) & 0x7fffffffffffffffULL) == 0x7ff0000000000000ULL; }
F3DS_HD bool m_signbit(double x) { return (m_bits(x) >> 63) != 0; }
F3DS_HD double m_copysign(double mag, double sgn) {
    return m_from_bits((m_bits(mag) & 0x7fffffffffffffffULL) | (m_bits(sgn) & 0x8000000000000000ULL));
}
F3DS_HD double m_nan() { return m_from_bits(0x7ff8000000000000ULL); }
F3DS_HD double m_inf() { return m_from_bits(0x7ff0000000000000ULL); }
F3DS_HD bool m_isfinitef(float x) { return (m_bitsf(x) & 0x7f800000u) != 0x7f800000u; }

// 2^k as a double for -1022 <= k <= 1023
F3DS_HD double m_pow2(int k) { return m_from_bits((uint64_t)(k + 1023) << 52); }

// Fused multiply-add is an IEEE-754 basic operation with one rounding: v_fma_f64 on the device, vfmadd (or glibc's exact
// fma) on the host give the same bits.  Everything below is written with explicit fma -- never contracted by the compiler
// (-ffp-contract=off) -- because on the GPU these functions sit on the serial path of the merge loop and a wave issues
// one f64 instruction every ~4.8 cycles whatever the number of active lanes (tools/ubench): instruction count is latency.
// Round 1's versions (separate multiply and add, 16-term series, 64-bit integer conversions) cost 500-1100 cycles each.
F3DS_HD double m_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Polynomials are evaluated with Estrin's scheme (independent sub-sums combined with z^2, z^4, z^8).  The evaluation
// order below is part of the bit-exact host/device contract.
F3DS_HD double m_estrin8(double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7, double z) {
    const double z2 = z * z, z4 = z2 * z2;
    const double a0 = m_fma(c1, z, c0), a1 = m_fma(c3, z, c2), a2 = m_fma(c5, z, c4), a3 = m_fma(c7, z, c6);
    const double b0 = m_fma(a1, z2, a0), b1 = m_fma(a3, z2, a2);
    return m_fma(b1, z4, b0);
}
F3DS_HD double m_estrin12(double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7, double c8, double c9,
                          double c10, double c11, double z) {
    const double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double a0 = m_fma(c1, z, c0), a1 = m_fma(c3, z, c2), a2 = m_fma(c5, z, c4), a3 = m_fma(c7, z, c6), a4 = m_fma(c9, z, c8), a5 = m_fma(c11, z, c10);
    const double b0 = m_fma(a1, z2, a0), b1 = m_fma(a3, z2, a2), b2 = m_fma(a5, z2, a4);
    return m_fma(b2, z8, m_fma(b1, z4, b0));
}
F3DS_HD double m_estrin14(double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7, double c8, double c9,
                          double c10, double c11, double c12, double c13, double z) {
    const double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double a0 = m_fma(c1, z, c0), a1 = m_fma(c3, z, c2), a2 = m_fma(c5, z, c4), a3 = m_fma(c7, z, c6);
    const double a4 = m_fma(c9, z, c8), a5 = m_fma(c11, z, c10), a6 = m_fma(c13, z, c12);
    const double b0 = m_fma(a1, z2, a0), b1 = m_fma(a3, z2, a2), b2 = m_fma(a5, z2, a4);
    const double d0 = m_fma(b1, z4, b0), d1 = m_fma(a6, z4, b2);
    return m_fma(d1, z8, d0);
}
// ---- where the f64 constants come from ---------------------------------------------------------
// Every f64 constant of the functions below sits in one table, M_TABLE, and the functions read it through a provider `c(index)`:
//   m_lit  (the default)  the constant itself: the compiler sees a literal, as if it were written in place;
//   m_tab  a pointer to a copy of the table (the merge loop keeps one in LDS).
// Why: gfx9 has no 64-bit literals.  A double that is not one of the inline constants costs two s_mov_b32, and a second constant in the same
// instruction (fma(c1, z, c0): every first-level term of Estrin's scheme) two v_mov_b32 more -- five instructions for one fma, on a wave
// that issues one instruction per ~4.6 cycles; the SGPRs they occupy push the loop's own state into spills.  Of the 549 instructions the merge loop
// spent on one rgb -> Lab evaluation, 205 were such moves.  Two adjacent table entries arrive with one ds_read_b128.  Same values either way.
enum {
    MC_EXP_P = 0,            // 14: Taylor coefficients 1/k!
    MC_INV_LN2 = 14, MC_MAGIC, MC_EXP_LN2_HI, MC_EXP_LN2_LO, MC_EXP_LO, MC_EXP_HI,
    MC_LOG_P = 20,           // 12: 1/(2k+3)
    MC_LOG_LN2_HI = 32, MC_LOG_LN2_LO, MC_SQRT2, MC_TWO54, MC_TINY, MC_TWO,
    MC_SIN_P = 38,           // 8
    MC_COS_P = 46,           // 8
    MC_TWO_OVER_PI = 54, MC_TWO30, MC_PIO2_HI, MC_PIO2_LO,
    MC_ATAN_P = 58,          // 14
    MC_ATAN_HALF_HI = 72, MC_ATAN_HALF_LO, MC_ATAN_ONE_HI, MC_ATAN_ONE_LO, MC_PI_HI, MC_PI_LO, MC_Q25, MC_Q75,
    MC_THIRD = 80, MC_THREE, MC_GAMMA_EXP, MC_PAD83,
    // lab_ciede00 (f3ds_numerics.h)
    MC_CIE_PI = 84, MC_CIE_2PI, MC_CIE_25_7, MC_CIE_50, MC_CIE_017, MC_CIE_024, MC_CIE_032, MC_CIE_020, MC_CIE_PI_6, MC_CIE_PI_30, MC_CIE_63PI_180, MC_CIE_180_PI,
    MC_CIE_275 = 96, MC_CIE_25, MC_CIE_30PI_180, MC_CIE_20, MC_CIE_0015, MC_CIE_0045, MC_CIE_3, MC_CIE_4,
    MC_COUNT = 104
};
#define F3DS_M_TABLE_INIT { \
    1.0, 1.0, 0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0, 1.0 / 40320.0, 1.0 / 362880.0, 1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0, \
    1.4426950408889634, 0x1.8p52, 0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56, -745.2, 709.782712893384, \
    1.0 / 3.0, 1.0 / 5.0, 1.0 / 7.0, 1.0 / 9.0, 1.0 / 11.0, 1.0 / 13.0, 1.0 / 15.0, 1.0 / 17.0, 1.0 / 19.0, 1.0 / 21.0, 1.0 / 23.0, 1.0 / 25.0, \
    0x1.62e4200000000p-1, 0x1.fdf473de6af28p-22, 1.4142135623730951, 0x1p54, 0x1p-1022, 2.0, \
    -1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0, -1.0 / 39916800.0, 1.0 / 6227020800.0, -1.0 / 1307674368000.0, 1.0 / 355687428096000.0, \
    1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0, -1.0 / 3628800.0, 1.0 / 479001600.0, -1.0 / 87178291200.0, 1.0 / 20922789888000.0, -1.0 / 6402373705728000.0, \
    0.6366197723675814, 0x1p30, 0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54, \
    -1.0 / 3.0, 1.0 / 5.0, -1.0 / 7.0, 1.0 / 9.0, -1.0 / 11.0, 1.0 / 13.0, -1.0 / 15.0, 1.0 / 17.0, -1.0 / 19.0, 1.0 / 21.0, -1.0 / 23.0, 1.0 / 25.0, -1.0 / 27.0, 1.0 / 29.0, \
    0x1.dac670561bb4fp-2, 0x1.a2b7f222f65e2p-56, 0x1.921fb54442d18p-1, 0x1.1a62633145c07p-55, 0x1.921fb54442d18p+1, 0x1.1a62633145c07p-53, 0.25, 0.75, \
    0x1.5555555555555p-2, 3.0, 2.4, 0.0, \
    3.14159265358979323846, 2.0 * 3.14159265358979323846, 6103515625.0, 50.0, 0.17, 0.24, 0.32, 0.20, 3.14159265358979323846 / 6.0, 3.14159265358979323846 / 30.0, \
    63.0 * 3.14159265358979323846 / 180.0, 180.0 / 3.14159265358979323846, \
    275.0, 25.0, 30.0 * 3.14159265358979323846 / 180.0, 20.0, 0.015, 0.045, 3.0, 4.0 }
struct m_lit {
    F3DS_HD double operator()(int i) const { constexpr double t[MC_COUNT] = F3DS_M_TABLE_INIT; return t[i]; }
};
struct m_tab {
    const double* t;
    F3DS_HD double operator()(int i) const { return t[i]; }
};
// the table as data (what a kernel copies into LDS for m_tab)
F3DS_HD void m_table_fill(double* dst, int first, int step) {
    constexpr double t[MC_COUNT] = F3DS_M_TABLE_INIT;
    for (int i = first; i < MC_COUNT; i += step) dst[i] = t[i];
}

template <class K> F3DS_HD double m_poly8(K c, int at, double z) {
    return m_estrin8(c(at), c(at + 1), c(at + 2), c(at + 3), c(at + 4), c(at + 5), c(at + 6), c(at + 7), z);
}
template <class K> F3DS_HD double m_poly12(K c, int at, double z) {
    return m_estrin12(c(at), c(at + 1), c(at + 2), c(at + 3), c(at + 4), c(at + 5), c(at + 6), c(at + 7), c(at + 8), c(at + 9), c(at + 10), c(at + 11), z);
}
template <class K> F3DS_HD double m_poly14(K c, int at, double z) {
    return m_estrin14(c(at), c(at + 1), c(at + 2), c(at + 3), c(at + 4), c(at + 5), c(at + 6), c(at + 7), c(at + 8), c(at + 9), c(at + 10), c(at + 11), c(at + 12), c(at + 13), z);
}

// round to nearest integer (ties to even) for |t| < 2^51 without a 64-bit integer conversion; *q = the integer's low bits
F3DS_HD double m_rint_small(double t, int* q, double MAGIC = 0x1.8p52) {
    const double s = t + MAGIC;
    *q = (int)(uint32_t)m_bits(s);
    return s - MAGIC;
}

// ---- exp -------------------------------------------------------------------------------
// No branches: the main path is evaluated for every argument (nothing traps; for NaN / out-of-range arguments it produces bits that are
// thrown away) and the special results are selected afterwards.  A lone wave pays ~30 cycles for every `if (...) return` region
// (exec-mask bookkeeping + a taken branch), and divergent lanes pay for both sides anyway (tools/ubench/ubench_consts.hip: 380 -> 258 cycles).
F3DS_HD double m_pow2u(int k) { return m_from_bits((uint64_t)((uint32_t)k + 1023u) << 52); }      // m_pow2 for any int (wraps instead of overflowing; same bits for -1022..1023)
template <class K = m_lit> F3DS_HD double m_exp(double x, K c = K()) {
    int k;
    const double kd = m_rint_small(x * c(MC_INV_LN2), &k, c(MC_MAGIC));
    const double r = m_fma(-kd, c(MC_EXP_LN2_LO), m_fma(-kd, c(MC_EXP_LN2_HI), x));    // |r| <= ~0.3466
    // Taylor series, degree 13 (r^14/14! < 5e-18)
    const double p = m_poly14(c, MC_EXP_P, r);
    // p * 2^k in two exact steps: 2^k1 with |k1| <= 1000, then 2^(k - k1) (= 1.0 unless the result is near the ends of the range)
    const int k1 = k > 1000 ? 1000 : (k < -1000 ? -1000 : k);
    double v = (p * m_pow2u(k1)) * m_pow2u(k - k1);
    v = x < c(MC_EXP_LO) ? 0.0 : v;
    v = !(x <= c(MC_EXP_HI)) ? (x != x ? x : m_inf()) : v;      // NaN or overflow
    return v;
}

// ---- log -------------------------------------------------------------------------------
template <class K = m_lit> F3DS_HD double m_log(double x0, K c = K()) {
    const bool sub = x0 < c(MC_TINY);                  // subnormal (or not positive: selected away below)
    const double x = sub ? x0 * c(MC_TWO54) : x0;
    const uint64_t u = m_bits(x);
    int e = (sub ? -54 : 0) + ((int)(u >> 52) - 1023);
    double m = m_from_bits((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);   // [1,2)
    const bool big = m > c(MC_SQRT2);
    m = big ? m * 0.5 : m; e = big ? e + 1 : e;                                     // [0.7071,1.4142]
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    // 2*atanh(s) = 2s * (1 + z/3 + z^2/5 + ...),  z <= 0.0295: 12 terms of 1/(2k+3) (z^12/27 < 2e-20)
    const double p = m_poly12(c, MC_LOG_P, z);
    const double ed = (double)e;
    const double two_s = 2.0 * s;
    double r = m_fma(two_s, z * p, ed * c(MC_LOG_LN2_LO));      // LN2_HI has 20 significant bits: ed * LN2_HI is exact
    r = r + two_s;
    double v = m_fma(ed, c(MC_LOG_LN2_HI), r);
    v = m_isinf(x0) ? x0 : v;
    v = !(x0 > 0.0) ? (x0 == 0.0 ? -m_inf() : m_nan()) : v;           // 0, negative, NaN
    return v;
}

// ---- sin / cos -------------------------------------------------------------------------
// Reduction by pi/2 in two fused steps (pi/2 to ~107 bits); exact enough for |x| < ~1e5, which covers every argument
// of the path (hue angles in [0, 4*pi], theta in [0, pi/3]).  |x| >= 2^30 is outside the supported range (NaN).
template <class K = m_lit> F3DS_HD double m_sin_kernel(double r, K c = K()) {
    const double z = r * r;
    const double p = m_poly8(c, MC_SIN_P, z);
    return m_fma(r, z * p, r);
}
template <class K = m_lit> F3DS_HD double m_cos_kernel(double r, K c = K()) {
    const double z = r * r;
    const double p = m_poly8(c, MC_COS_P, z);
    // 1 - z/2 + z^2 p: the large terms first, exactly like the classic kernel (1 - z/2 is exact to one rounding)
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * (z * p));
}
template <class K = m_lit> F3DS_HD int m_rem_pio2(double x, double* r, K c = K()) {
    int q;
    const double nd = m_rint_small(x * c(MC_TWO_OVER_PI), &q, c(MC_MAGIC));
    *r = m_fma(-nd, c(MC_PIO2_LO), m_fma(-nd, c(MC_PIO2_HI), x));
    return q & 3;
}
// Both kernels are evaluated and one is selected: lanes with different quadrants would run both anyway, and a lone wave saves the branch.
template <class K = m_lit> F3DS_HD double m_sin(double x, K c = K()) {
    double r; const int q = m_rem_pio2(x, &r, c);
    const double sk = m_sin_kernel(r, c), ck = m_cos_kernel(r, c);
    const double v = (q & 1) ? ck : sk;
    const double sv = (q & 2) ? -v : v;
    return !(m_abs(x) < c(MC_TWO30)) ? m_nan() : sv;          // NaN, infinity, outside the supported range (documented)
}
template <class K = m_lit> F3DS_HD double m_cos(double x, K c = K()) {
    double r; const int q = m_rem_pio2(x, &r, c);
    const double sk = m_sin_kernel(r, c), ck = m_cos_kernel(r, c);
    const double v = (q & 1) ? sk : ck;
    const double sv = ((q + 1) & 2) ? -v : v;
    return !(m_abs(x) < c(MC_TWO30)) ? m_nan() : sv;
}

// ---- atan2 -----------------------------------------------------------------------------
// atan(t) for t in [0,1]: atan(t) = atan(c) + atan((t-c)/(1+t*c)), c in {0, 1/2, 1}
template <class K = m_lit> F3DS_HD double m_atan01(double t, K c = K()) {
    // branch-free choice of c (a wave evaluates this for many lanes at once: every taken branch would be executed by all)
    const bool lo_r = t < c(MC_Q25), mid_r = t < c(MC_Q75);
    const double cc = lo_r ? 0.0 : (mid_r ? 0.5 : 1.0);
    const double hi = lo_r ? 0.0 : (mid_r ? c(MC_ATAN_HALF_HI) : c(MC_ATAN_ONE_HI));
    const double lo = lo_r ? 0.0 : (mid_r ? c(MC_ATAN_HALF_LO) : c(MC_ATAN_ONE_LO));
    const double u = (t - cc) / m_fma(cc, t, 1.0);     // c = 0: t / 1 = t exactly
    const double z = u * u;                          // <= 0.0625
    // 14 terms: z^14/29 < 5e-19
    const double p = m_poly14(c, MC_ATAN_P, z);
    const double a = m_fma(u, z * p, u);
    return hi + (a + lo);
}
template <class K = m_lit> F3DS_HD double m_atan2(double y, double x, K c = K()) {
    const double PI_HI = c(MC_PI_HI), PI_LO = c(MC_PI_LO);
    const double PIO2_HI = c(MC_PIO2_HI), PIO2_LO = c(MC_PIO2_LO);
    const double ax = m_abs(x), ay = m_abs(y);
    double a;
    if (!(ax > 0.0 && ay > 0.0 && ax < m_inf() && ay < m_inf())) {      // zeros, infinities, NaN: the special cases of C99 F.9.1.4
        if (m_isnan(x) || m_isnan(y)) return m_nan();
        if (ay == 0.0) { a = m_signbit(x) ? PI_HI : 0.0; return m_copysign(a, y); }
        if (ax == 0.0) return m_copysign(PIO2_HI, y);
        if (m_isinf(ax) && m_isinf(ay)) {
            a = c(MC_ATAN_ONE_HI);
            if (m_signbit(x)) a = (PI_HI - a) + PI_LO;
            return m_copysign(a, y);
        }
        if (m_isinf(ax)) { a = m_signbit(x) ? PI_HI : 0.0; return m_copysign(a, y); }
        return m_copysign(PIO2_HI, y);
    }
    const bool steep = ay > ax;
    a = m_atan01((steep ? ax : ay) / (steep ? ay : ax), c);
    if (steep) a = (PIO2_HI - a) + PIO2_LO;
    if (m_signbit(x)) a = (PI_HI - a) + PI_LO;
    return m_copysign(a, y);
}

// ---- pow for positive base, cube root ---------------------------------------------------
template <class K = m_lit> F3DS_HD double m_pow_pos(double x, double y, K c = K()) {       // x > 0
    return m_exp(y * m_log(x, c), c);
}
template <class K = m_lit> F3DS_HD double m_cbrt_pos(double x, K c = K()) {                // x > 0
    double y = m_exp(m_log(x, c) * c(MC_THIRD), c);
    // one Newton step removes the exp/log rounding: y -= (y^3 - x) / (3 y^2)
    const double y2 = y * y;
    y = y - m_fma(y2, y, -x) / (c(MC_THREE) * y2);
    return y;
}

// ---- float wrappers (double evaluation, one final rounding) -----------------------------
template <class K = m_lit> F3DS_HD float m_logf(float x, K c = K()) { return (float)m_log((double)x, c); }
template <class K = m_lit> F3DS_HD float m_atan2f(float y, float x, K c = K()) { return (float)m_atan2((double)y, (double)x, c); }
template <class K = m_lit> F3DS_HD float m_cosf(float x, K c = K()) { return (float)m_cos((double)x, c); }
template <class K = m_lit> F3DS_HD float m_sinf(float x, K c = K()) { return (float)m_sin((double)x, c); }

}  // namespace f3ds
#endif  // F3DS_MATH_H_
