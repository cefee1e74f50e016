// f3ds_math.h -- transcendental functions built from IEEE-754 basic operations only.
//
// Why this exists: the segmentation path takes a logarithm before voxel keys are
// generated (single-camera transform) and uses atan2/cos/sin/exp inside the merge
// distance.  libm (host) and ocml (device) do not return the same last bit for those,
// and one flipped bit can move a point to the neighbouring voxel or reorder two merges.
// Every function below uses only + - * / sqrt, integer bit moves and comparisons, so
// the same source gives the same bits from g++ (x86-64, SSE2) and from hipcc (gfx950),
// provided both are compiled with -ffp-contract=off (the build scripts do that).
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


// Polynomials are evaluated with Estrin's scheme (independent sub-sums combined with z^2, z^4,
// z^8) rather than Horner's: on the GPU a dependent f64 operation costs ~14-16 cycles while an
// independent one issues every 4, and these functions sit on the serial path of the merge loop.
// The evaluation order below is part of the bit-exact host/device contract.
F3DS_HD double m_estrin8(double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7, double z) {
    const double z2 = z * z, z4 = z2 * z2;
    const double a0 = c0 + c1 * z, a1 = c2 + c3 * z, a2 = c4 + c5 * z, a3 = c6 + c7 * z;
    const double b0 = a0 + a1 * z2, b1 = a2 + a3 * z2;
    return b0 + b1 * z4;
}
F3DS_HD double m_estrin16(double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7, double c8, double c9,
                          double c10, double c11, double c12, double c13, double c14, double c15, double z) {
    const double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double a0 = c0 + c1 * z, a1 = c2 + c3 * z, a2 = c4 + c5 * z, a3 = c6 + c7 * z;
    const double a4 = c8 + c9 * z, a5 = c10 + c11 * z, a6 = c12 + c13 * z, a7 = c14 + c15 * z;
    const double b0 = a0 + a1 * z2, b1 = a2 + a3 * z2, b2 = a4 + a5 * z2, b3 = a6 + a7 * z2;
    const double d0 = b0 + b1 * z4, d1 = b2 + b3 * z4;
    return d0 + d1 * z8;
}

// ---- exp -------------------------------------------------------------------------------
F3DS_HD double m_exp(double x) {
    if (m_isnan(x)) return x;
    if (x > 709.782712893384) return m_inf();
    if (x < -745.2) return 0.0;
    const double INV_LN2 = 1.4426950408889634;
    const double LN2_HI = 0x1.62e4200000000p-1;   // 20 significant bits: k*LN2_HI is exact
    const double LN2_LO = 0x1.fdf473de6af28p-22;
    double t = x * INV_LN2;
    int k = (int)(t + (t < 0.0 ? -0.5 : 0.5));
    double kd = (double)k;
    double r = (x - kd * LN2_HI) - kd * LN2_LO;    // |r| <= ~0.3466
    // Taylor series, degree 15 (r^16/16! < 1e-20)
    const double p = m_estrin16(1.0, 1.0, 0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0, 1.0 / 40320.0, 1.0 / 362880.0,
                                1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0, 1.0 / 87178291200.0,
                                1.0 / 1307674368000.0, r);
    if (k > 1000) return (p * m_pow2(1000)) * m_pow2(k - 1000);
    if (k < -1000) return (p * m_pow2(-1000)) * m_pow2(k + 1000);
    return p * m_pow2(k);
}

// ---- log -------------------------------------------------------------------------------
F3DS_HD double m_log(double x) {
    if (m_isnan(x)) return x;
    if (x < 0.0) return m_nan();
    if (x == 0.0) return -m_inf();
    if (m_isinf(x)) return x;
    int e = 0;
    if (x < 0x1p-1022) { x = x * 0x1p54; e = -54; }   // subnormal
    uint64_t u = m_bits(x);
    e += (int)(u >> 52) - 1023;
    double m = m_from_bits((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);   // [1,2)
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }                       // [0.7071,1.4142]
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    // 2*atanh(s) = 2s * (1 + z/3 + z^2/5 + ...),  z <= 0.0295: 16 terms of 1/(2k+3)
    const double p = m_estrin16(1.0 / 3.0, 1.0 / 5.0, 1.0 / 7.0, 1.0 / 9.0, 1.0 / 11.0, 1.0 / 13.0, 1.0 / 15.0, 1.0 / 17.0, 1.0 / 19.0, 1.0 / 21.0,
                                1.0 / 23.0, 1.0 / 25.0, 1.0 / 27.0, 1.0 / 29.0, 1.0 / 31.0, 1.0 / 33.0, z);
    const double LN2_HI = 0x1.62e4200000000p-1;
    const double LN2_LO = 0x1.fdf473de6af28p-22;
    double ed = (double)e;
    double two_s = 2.0 * s;
    double r = ed * LN2_LO + two_s * (z * p);
    r = r + two_s;
    r = r + ed * LN2_HI;
    return r;
}

// ---- sin / cos -------------------------------------------------------------------------
// Cody-Waite reduction by pi/2 in three pieces; exact enough for |x| < ~1e5, which covers
// every argument of the path (hue angles in [0, 4*pi], theta in [0, pi/3]).
F3DS_HD double m_sin_kernel(double r) {
    const double z = r * r;
    const double p = m_estrin8(-1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0, -1.0 / 39916800.0, 1.0 / 6227020800.0,
                               -1.0 / 1307674368000.0, 1.0 / 355687428096000.0, z);
    return r + r * (z * p);
}
F3DS_HD double m_cos_kernel(double r) {
    const double z = r * r;
    const double p = m_estrin16(-0.5, 1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0, -1.0 / 3628800.0, 1.0 / 479001600.0, -1.0 / 87178291200.0,
                                1.0 / 20922789888000.0, -1.0 / 6402373705728000.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, z);
    return 1.0 + z * p;
}
F3DS_HD int m_rem_pio2(double x, double* r) {
    const double TWO_OVER_PI = 0.6366197723675814;
    const double PIO2_1 = 0x1.921fb54400000p+0;      // 33 bits
    const double PIO2_2 = 0x1.0b4611a600000p-34;     // next 33 bits
    const double PIO2_3 = 0x1.3198a2e037073p-69;
    double t = x * TWO_OVER_PI;
    double nd = (double)(long long)(t + (t < 0.0 ? -0.5 : 0.5));
    *r = ((x - nd * PIO2_1) - nd * PIO2_2) - nd * PIO2_3;
    return (int)((long long)nd & 3);
}
F3DS_HD double m_sin(double x) {
    if (m_isnan(x) || m_isinf(x)) return m_nan();
    if (m_abs(x) <= 0.7853981633974483) return m_sin_kernel(x);
    if (m_abs(x) > 1.0e15) return m_nan();          // outside the supported range (documented)
    double r; int q = m_rem_pio2(x, &r);
    switch (q) {
        case 0: return m_sin_kernel(r);
        case 1: return m_cos_kernel(r);
        case 2: return -m_sin_kernel(r);
        default: return -m_cos_kernel(r);
    }
}
F3DS_HD double m_cos(double x) {
    if (m_isnan(x) || m_isinf(x)) return m_nan();
    if (m_abs(x) <= 0.7853981633974483) return m_cos_kernel(x);
    if (m_abs(x) > 1.0e15) return m_nan();
    double r; int q = m_rem_pio2(x, &r);
    switch (q) {
        case 0: return m_cos_kernel(r);
        case 1: return -m_sin_kernel(r);
        case 2: return -m_cos_kernel(r);
        default: return m_sin_kernel(r);
    }
}

// ---- atan2 -----------------------------------------------------------------------------
// atan(t) for t in [0,1]: atan(t) = atan(c) + atan((t-c)/(1+t*c)), c in {0, 1/2, 1}
F3DS_HD double m_atan01(double t) {
    double hi, lo, u;
    if (t < 0.25) { hi = 0.0; lo = 0.0; u = t; }
    else if (t < 0.75) { hi = 0x1.dac670561bb4fp-2; lo = 0x1.a2b7f222f65e2p-56; u = (t - 0.5) / (1.0 + 0.5 * t); }
    else { hi = 0x1.921fb54442d18p-1; lo = 0x1.1a62633145c07p-55; u = (t - 1.0) / (1.0 + t); }
    const double z = u * u;                          // <= 0.0625
    const double p = m_estrin16(-1.0 / 3.0, 1.0 / 5.0, -1.0 / 7.0, 1.0 / 9.0, -1.0 / 11.0, 1.0 / 13.0, -1.0 / 15.0, 1.0 / 17.0, -1.0 / 19.0,
                                1.0 / 21.0, -1.0 / 23.0, 1.0 / 25.0, -1.0 / 27.0, 1.0 / 29.0, -1.0 / 31.0, 1.0 / 33.0, z);
    double a = u + u * (z * p);
    return hi + (a + lo);
}
F3DS_HD double m_atan2(double y, double x) {
    if (m_isnan(x) || m_isnan(y)) return m_nan();
    const double PI_HI = 0x1.921fb54442d18p+1, PI_LO = 0x1.1a62633145c07p-53;
    const double PIO2_HI = 0x1.921fb54442d18p+0, PIO2_LO = 0x1.1a62633145c07p-54;
    double ax = m_abs(x), ay = m_abs(y);
    double a;
    if (ay == 0.0) {
        a = m_signbit(x) ? PI_HI : 0.0;
        return m_copysign(a, y);
    }
    if (ax == 0.0) return m_copysign(PIO2_HI, y);
    if (m_isinf(ax) && m_isinf(ay)) {
        a = 0x1.921fb54442d18p-1;
        if (m_signbit(x)) a = (PI_HI - a) + PI_LO;
        return m_copysign(a, y);
    }
    if (m_isinf(ax)) { a = m_signbit(x) ? PI_HI : 0.0; return m_copysign(a, y); }
    if (m_isinf(ay)) return m_copysign(PIO2_HI, y);
    if (ax >= ay) a = m_atan01(ay / ax);
    else a = (PIO2_HI - m_atan01(ax / ay)) + PIO2_LO;
    if (m_signbit(x)) a = (PI_HI - a) + PI_LO;
    return m_copysign(a, y);
}

// ---- pow for positive base, cube root ---------------------------------------------------
F3DS_HD double m_pow_pos(double x, double y) {       // x > 0
    return m_exp(y * m_log(x));
}
F3DS_HD double m_cbrt_pos(double x) {                // x > 0
    double y = m_exp(m_log(x) / 3.0);
    // one Newton step removes the exp/log rounding: y -= (y^3 - x) / (3 y^2)
    double y2 = y * y;
    y = y - (y2 * y - x) / (3.0 * y2);
    return y;
}

// ---- float wrappers (double evaluation, one final rounding) -----------------------------
F3DS_HD float m_logf(float x) { return (float)m_log((double)x); }
F3DS_HD float m_atan2f(float y, float x) { return (float)m_atan2((double)y, (double)x); }
F3DS_HD float m_cosf(float x) { return (float)m_cos((double)x); }
F3DS_HD float m_sinf(float x) { return (float)m_sin((double)x); }

}  // namespace f3ds
#endif  // F3DS_MATH_H_
