// f3ds_numerics.h -- per-element arithmetic of the segmentation path, callable from HIP kernels
// (device) and from host code (CLI, CPU-side unit tests of this header).
//
// Every function states the reference arithmetic it reproduces (file:line into /root/reference,
// or the SURVEY.md appendix item for the PCL / OpenCV half that is not in the reference tree).
// Float evaluation order is part of the contract: compile with -ffp-contract=off.
#ifndef F3DS_NUMERICS_H_
#define F3DS_NUMERICS_H_

#include "f3ds_math.h"

namespace f3ds {

#define F3DS_FLT_MAX 3.402823466e+38f
#define F3DS_FLT_MIN 1.175494351e-38f
#define F3DS_FLT_EPS 1.192092896e-07f

F3DS_HD float n_sqrtf(float x) { return __builtin_sqrtf(x); }
F3DS_HD double n_sqrt(double x) { return __builtin_sqrt(x); }
F3DS_HD bool n_finite3(float x, float y, float z) { return m_isfinitef(x) && m_isfinitef(y) && m_isfinitef(z); }
F3DS_HD float n_nanf() { return m_from_bitsf(0x7fc00000u); }

// -------------------------------------------------------------------------------------------
// voxel grid  (SupervoxelClustering::transformFunction,
// OctreePointCloud::defineBoundingBox / getKeyBitSize / genOctreeKeyforPoint)
// -------------------------------------------------------------------------------------------
struct GridInfo {
    double min[3];
    double max[3];
    double res;
    int depth;
    unsigned max_key;
    int error;        // 0, or F3DS_ERR_DEPTH (-4)
    int empty;        // 1 when no point survived the bounding-box pass
};

// main()'s z<0 -> |z| (src/supervoxel_clustering.cpp:317-321) then the single-camera transform
F3DS_HD void n_prelude(float& z, int fold_negative_z) {
    if (fold_negative_z && z < 0.0f) z = m_absf(z);
}
F3DS_HD void n_transform(float& x, float& y, float& z, int use_transform) {
    if (use_transform) { x = x / z; y = y / z; z = m_logf(z); }
}

// cube side is a power of two voxels, centred on the data
F3DS_HD void n_key_bit_size(GridInfo& g) {
    const double eps = (double)F3DS_FLT_EPS;
    unsigned mk = 2u;
    for (int a = 0; a < 3; ++a) {
        unsigned k = (unsigned)__builtin_ceil((g.max[a] - g.min[a] - eps) / g.res);
        if (k > mk) mk = k;
    }
    double lg = m_log((double)mk) / m_log(2.0) - eps;
    unsigned d = (unsigned)__builtin_ceil(lg);
    if (d > 32u) d = 32u;
    if (d > 21u) { g.error = -4; g.depth = (int)d; return; }
    g.depth = (int)d;
    g.max_key = (1u << d) - 1u;
    double side = (double)(1u << d) * g.res;
    for (int a = 0; a < 3; ++a) {
        double over = (side - (g.max[a] - g.min[a])) / 2.0;
        if (over > eps) { g.min[a] -= over; g.max[a] += over; }
    }
}
F3DS_HD void n_grid_from_bbox(const float mn[3], const float mx[3], float voxel_res, GridInfo& g) {
    g.error = 0; g.empty = 0; g.depth = 0; g.max_key = 0;
    g.res = (double)voxel_res;
    for (int a = 0; a < 3; ++a) {
        double lo = (double)mn[a], hi = (double)mx[a];
        g.min[a] = lo < hi ? lo : hi;
        g.max[a] = lo < hi ? hi : lo;
    }
    n_key_bit_size(g);
}
// key of an (already folded) input point; finite original coordinates are the caller's check
F3DS_HD void n_point_key(const GridInfo& g, float x, float y, float z, int use_transform, unsigned key[3]) {
    n_transform(x, y, z, use_transform);
    if (!use_transform || n_finite3(x, y, z)) {
        // (unsigned)(a / res) without the division where that is provably the same number: a * (1 / res) is within 2.2e-16 of a / res relatively, the correctly rounded
        // quotient within 1.1e-16, keys are below 2^21 -- so whenever the product lies more than 1e-6 away from an integer, quotient and product truncate alike.  Only a
        // coordinate that close to a cell border takes the division (three f64 divisions per point were a third of the key kernel's instructions).
        const double inv = 1.0 / g.res;
        const float p[3] = {x, y, z};
        for (int a = 0; a < 3; ++a) {
            const double d = (double)p[a] - g.min[a];
            const double q1 = d * inv;
            const unsigned k1 = (unsigned)q1;
            const double f = q1 - (double)k1;
            key[a] = (f > 1e-6 && f < 1.0 - 1e-6 && q1 < 4194304.0) ? k1 : (unsigned)(d / g.res);
        }
    } else {
        key[0] = key[1] = key[2] = 0u;     // transformed point not finite -> default OctreeKey
    }
}
// depth-first leaf order of the octree = this code ascending (child = x<<2|y<<1|z per level)
// (bit b of x, y, z -> bits 3b + 2, 3b + 1, 3b, for b < depth: the coordinates' bits spread three apart with the usual mask-and-shift steps instead of a loop over the bits --
// 32-bit steps up to depth 10, i.e. every frame of a depth camera)
F3DS_HD uint32_t n_spread3_10(uint32_t v) {        // 10 bits
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
F3DS_HD uint64_t n_spread3_21(uint64_t v) {        // 21 bits
    v = (v | (v << 32)) & 0x001F00000000FFFFull;
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
F3DS_HD uint64_t n_morton(unsigned x, unsigned y, unsigned z, int depth) {
    if (depth <= 10) {
        const uint32_t m = (1u << depth) - 1u;
        return (uint64_t)((n_spread3_10(x & m) << 2) | (n_spread3_10(y & m) << 1) | n_spread3_10(z & m));
    }
    const uint64_t m = (1ull << depth) - 1ull;
    return (n_spread3_21(x & m) << 2) | (n_spread3_21(y & m) << 1) | n_spread3_21(z & m);
}
F3DS_HD void n_demorton(uint64_t c, int depth, unsigned key[3]) {
    unsigned x = 0, y = 0, z = 0;
    for (int b = depth - 1; b >= 0; --b) {
        unsigned d = (unsigned)(c >> (3 * b)) & 7u;
        x = (x << 1) | ((d >> 2) & 1u);
        y = (y << 1) | ((d >> 1) & 1u);
        z = (z << 1) | (d & 1u);
    }
    key[0] = x; key[1] = y; key[2] = z;
}
F3DS_HD uint64_t n_pack_key(unsigned x, unsigned y, unsigned z) {
    return ((uint64_t)x << 42) | ((uint64_t)y << 21) | (uint64_t)z;
}

// -------------------------------------------------------------------------------------------
// plane normal from the nine running sums (computeMeanAndCovarianceMatrix,
// solvePlaneParameters, eigen33, computeRoots, flipNormalTowardsViewpoint + caller's
// normal[3]=0; normalize()).  accu = {xx,xy,xz,yy,yz,zz,x,y,z} raw sums, count points.
// Eigen reduction orders: 3-vector a+(b+c); 4-vector (a+b)+(c+d).
// -------------------------------------------------------------------------------------------
F3DS_HD void n_roots2(float b, float c, float r[3]) {
    r[0] = 0.0f;
    float d = (float)((double)(b * b) - 4.0 * (double)c);
    if (d < 0.0f) d = 0.0f;
    float sd = n_sqrtf(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}
template <class K = m_lit> F3DS_HD void n_roots(float m00, float m01, float m02, float m11, float m12, float m22, float r[3], K mc = K()) {
    float c0 = m00 * m11 * m22 + 2.0f * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 - m22 * m01 * m01;
    float c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
    float c2 = m00 + m11 + m22;
    if (m_absf(c0) < F3DS_FLT_EPS) { n_roots2(c2, c1, r); return; }
    const float inv3 = (float)(1.0 / 3.0);
    const float sqrt3 = 1.7320508075688772f;          // sqrtf(3.0f)
    float c2_3 = c2 * inv3;
    float a_3 = (c1 - c2 * c2_3) * inv3;
    if (a_3 > 0.0f) a_3 = 0.0f;
    float half_b = 0.5f * (c0 + c2_3 * (2.0f * c2_3 * c2_3 - c1));
    float q = half_b * half_b + a_3 * a_3 * a_3;
    if (q > 0.0f) q = 0.0f;
    float rho = n_sqrtf(-a_3);
    float theta = m_atan2f(n_sqrtf(-q), half_b, mc) * inv3;
    float ct = m_cosf(theta, mc), st = m_sinf(theta, mc);
    r[0] = c2_3 + 2.0f * rho * ct;
    r[1] = c2_3 - rho * (ct + sqrt3 * st);
    r[2] = c2_3 - rho * (ct - sqrt3 * st);
    float t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    if (r[1] >= r[2]) {
        t = r[1]; r[1] = r[2]; r[2] = t;
        if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    }
    if (r[0] <= 0.0f) n_roots2(c2, c1, r);
}
F3DS_HD float n_sum3(float a, float b, float c) { return a + (b + c); }
F3DS_HD void n_cross(float a0, float a1, float a2, float b0, float b1, float b2, float o[3]) {
    o[0] = a1 * b2 - a2 * b1;
    o[1] = a2 * b0 - a0 * b2;
    o[2] = a0 * b1 - a1 * b0;
}
// out: unit normal (w = 0) flipped towards the origin as seen from view_point
F3DS_HD void n_plane_normal(const float accu_raw[9], unsigned count, const float view_point[3], float out[4]) {
    float nx, ny, nz, nw;
    if (count < 3u) {
        nx = ny = nz = nw = n_nanf();
    } else {
        float cnt = (float)count;
        float a[9];
        for (int i = 0; i < 9; ++i) a[i] = accu_raw[i] / cnt;
        float c00 = a[0] - a[6] * a[6], c01 = a[1] - a[6] * a[7], c02 = a[2] - a[6] * a[8];
        float c11 = a[3] - a[7] * a[7], c12 = a[4] - a[7] * a[8], c22 = a[5] - a[8] * a[8];
        float scale = m_absf(c00);
        float t;
        t = m_absf(c01); if (t > scale) scale = t;
        t = m_absf(c02); if (t > scale) scale = t;
        t = m_absf(c11); if (t > scale) scale = t;
        t = m_absf(c12); if (t > scale) scale = t;
        t = m_absf(c22); if (t > scale) scale = t;
        // NaN handling of cwiseAbs().maxCoeff(): comparisons with NaN are false, as above
        if (scale <= F3DS_FLT_MIN) scale = 1.0f;
        float m00 = c00 / scale, m01 = c01 / scale, m02 = c02 / scale, m11 = c11 / scale, m12 = c12 / scale, m22 = c22 / scale;
        float r[3];
        n_roots(m00, m01, m02, m11, m12, m22, r);
        m00 -= r[0]; m11 -= r[0]; m22 -= r[0];
        float v1[3], v2[3], v3[3];
        n_cross(m00, m01, m02, m01, m11, m12, v1);      // row0 x row1
        n_cross(m00, m01, m02, m02, m12, m22, v2);      // row0 x row2
        n_cross(m01, m11, m12, m02, m12, m22, v3);      // row1 x row2
        float l1 = n_sum3(v1[0] * v1[0], v1[1] * v1[1], v1[2] * v1[2]);
        float l2 = n_sum3(v2[0] * v2[0], v2[1] * v2[1], v2[2] * v2[2]);
        float l3 = n_sum3(v3[0] * v3[0], v3[1] * v3[1], v3[2] * v3[2]);
        const float* v; float l;
        if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
        else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
        else { v = v3; l = l3; }
        float s = n_sqrtf(l);
        nx = v[0] / s; ny = v[1] / s; nz = v[2] / s;
        // Hessian component; only its NaN-ness can matter (0 * NaN in the flip test)
        nw = -1.0f * ((nx * a[6] + ny * a[7]) + (nz * a[8] + 0.0f * 1.0f));
    }
    // flipNormalTowardsViewpoint(point, 0, 0, 0, n): vp = (0-px, 0-py, 0-pz, 0)
    float cos_theta = ((0.0f - view_point[0]) * nx + (0.0f - view_point[1]) * ny) + ((0.0f - view_point[2]) * nz + 0.0f * nw);
    if (cos_theta < 0.0f) { nx *= -1.0f; ny *= -1.0f; nz *= -1.0f; }
    // normal[3] = 0; normalize()  (Eigen 3.3: only when squaredNorm > 0)
    float z = (nx * nx + ny * ny) + (nz * nz + 0.0f);
    if (z > 0.0f) { float s = n_sqrtf(z); nx /= s; ny /= s; nz /= s; }
    out[0] = nx; out[1] = ny; out[2] = nz; out[3] = 0.0f;
}

// -------------------------------------------------------------------------------------------
// VCCS feature distance (SupervoxelClustering::voxelDataDistance).
// feature rows are 12 floats: xyz[0..2] rgb[3..5] normal[6..8] (w = 0) pad[9..11]
// -------------------------------------------------------------------------------------------
F3DS_HD float n_voxel_distance(const float* c, const float* v, float seed_res, float w_normal, float w_color, float w_spatial) {
    float dx = c[0] - v[0], dy = c[1] - v[1], dz = c[2] - v[2];
    float spatial = n_sqrtf(n_sum3(dx * dx, dy * dy, dz * dz)) / seed_res;
    float er = c[3] - v[3], eg = c[4] - v[4], eb = c[5] - v[5];
    float color = n_sqrtf(n_sum3(er * er, eg * eg, eb * eb)) / 255.0f;
    float dot = (c[6] * v[6] + c[7] * v[7]) + (c[8] * v[8] + 0.0f * 0.0f);
    float cosn = 1.0f - m_absf(dot);
    return cosn * w_normal + color * w_color + spatial * w_spatial;
}

// -------------------------------------------------------------------------------------------
// colour metrics (src/color_utilities.cpp)
// -------------------------------------------------------------------------------------------
#define F3DS_RGB_RANGE 441.672943f   // include/supervoxel_clustering/color_utilities.h:62
#define F3DS_LAB_RANGE 137.3607f     // include/supervoxel_clustering/color_utilities.h:63

// rgb2lab (:151-160): /255 then cv::cvtColor(COLOR_RGB2Lab) on a float pixel; analytic sRGB/D65
// form (OpenCV is not vendored: unpinned)
template <class K = m_lit> F3DS_HD float n_lab_f(float t, K mc = K()) {
    return t > 0.008856f ? (float)m_cbrt_pos((double)t, mc) : 7.787f * t + 16.0f / 116.0f;
}
F3DS_HD void n_rgb2lab(const float rgb[3], float lab[3]) {
    float c[3];
    for (int i = 0; i < 3; ++i) {
        float v = rgb[i] / 255;
        c[i] = v <= 0.04045f ? v / 12.92f : (float)m_pow_pos((double)((v + 0.055f) / 1.055f), 2.4);
    }
    float X = (c[0] * 0.412453f + c[1] * 0.357580f + c[2] * 0.180423f) / 0.950456f;
    float Y = (c[0] * 0.212671f + c[1] * 0.715160f + c[2] * 0.072169f);
    float Z = (c[0] * 0.019334f + c[1] * 0.119193f + c[2] * 0.950227f) / 1.088754f;
    float fx = n_lab_f(X), fy = n_lab_f(Y), fz = n_lab_f(Z);
    lab[0] = Y > 0.008856f ? 116.0f * fy - 16.0f : 903.3f * Y;
    lab[1] = 500.0f * (fx - fy);
    lab[2] = 200.0f * (fy - fz);
}

// lab_ciede00 (:190-294) with kL = kC = kH = 1: float inputs, double intermediates, float result
F3DS_HD double n_pow7(double x) { double x2 = x * x; double x4 = x2 * x2; return (x4 * x2) * x; }
template <class K = m_lit> F3DS_HD float n_ciede00(const float lab1[3], const float lab2[3], K mc = K()) {
    const double PI = mc(MC_CIE_PI), TWO_PI = mc(MC_CIE_2PI);          // (2.0 * PI is exact: the same double as the table's)
    const double P25_7 = mc(MC_CIE_25_7);
    float L1 = lab1[0], a1 = lab1[1], b1 = lab1[2];
    float L2 = lab2[0], a2 = lab2[1], b2 = lab2[2];
    double Cab1 = (double)n_sqrtf(a1 * a1 + b1 * b1);
    double Cab2 = (double)n_sqrtf(a2 * a2 + b2 * b2);
    double Cab = (Cab1 + Cab2) / 2.0;
    double Cab7 = n_pow7(Cab);
    double G = 0.5 * (1.0 - n_sqrt(Cab7 / (Cab7 + P25_7)));
    double ap1 = (1.0 + G) * (double)a1;
    double ap2 = (1.0 + G) * (double)a2;
    double Cp1 = n_sqrt(ap1 * ap1 + (double)(b1 * b1));
    double Cp2 = n_sqrt(ap2 * ap2 + (double)(b2 * b2));
    double Cp_prod = Cp2 * Cp1;
    double hp1 = 0.0;
    if ((m_abs(ap1) + (double)m_absf(b1)) != 0.0) {
        hp1 = m_atan2((double)b1, ap1, mc);
        if (hp1 < 0) hp1 += TWO_PI;
    }
    double hp2 = 0.0;
    if ((m_abs(ap2) + (double)m_absf(b2)) != 0.0) {
        hp2 = m_atan2((double)b2, ap2, mc);
        if (hp2 < 0) hp2 += TWO_PI;
    }
    double dL = (double)(L2 - L1);
    double dC = Cp2 - Cp1;
    double dhp = hp2 - hp1;
    if (dhp > PI) dhp -= TWO_PI;
    else if (dhp < -PI) dhp += TWO_PI;
    if (Cp_prod == 0.0) dhp = 0.0;
    double dH = 2.0 * n_sqrt(Cp_prod) * m_sin(dhp / 2.0, mc);
    double Lp = (double)(L2 + L1) / 2.0;
    double Cp = (Cp1 + Cp2) / 2.0;
    double hp = (hp1 + hp2) / 2.0;
    if (m_abs(hp1 - hp2) > PI) hp -= PI;
    if (hp < 0) hp += TWO_PI;
    if (Cp_prod == 0.0) hp = hp1 + hp2;
    double Lpm502 = (Lp - mc(MC_CIE_50)) * (Lp - mc(MC_CIE_50));
    double T = 1.0 - mc(MC_CIE_017) * m_cos(hp - mc(MC_CIE_PI_6), mc) + mc(MC_CIE_024) * m_cos(2.0 * hp, mc) + mc(MC_CIE_032) * m_cos(mc(MC_CIE_3) * hp + mc(MC_CIE_PI_30), mc) -
               mc(MC_CIE_020) * m_cos(4.0 * hp - mc(MC_CIE_63PI_180), mc);
    double e = (mc(MC_CIE_180_PI) * hp - mc(MC_CIE_275)) / mc(MC_CIE_25);
    double dtheta = mc(MC_CIE_30PI_180) * m_exp(-(e * e), mc);
    double Cp7 = n_pow7(Cp);
    double Rc = 2.0 * n_sqrt(Cp7 / (Cp7 + P25_7));
    double kLSL = 1.0 * (1.0 + mc(MC_CIE_0015) * Lpm502 / n_sqrt(mc(MC_CIE_20) + Lpm502));
    double kLSC = 1.0 * (1.0 + mc(MC_CIE_0045) * Cp);
    double kHSH = 1.0 * (1.0 + mc(MC_CIE_0015) * Cp * T);
    double RT = -m_sin(2.0 * dtheta, mc) * Rc;
    double tL = dL / kLSL, tC = dC / kLSC, tH = dH / kHSH;
    return (float)n_sqrt(tL * tL + tC * tC + tH * tH + RT * tC * tH);
}
// rgb_eucl (:304-319): std::pow(float,int) squares in double, the result is stored to float
F3DS_HD float n_rgb_eucl(const float a[3], const float b[3]) {
    float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    float rd = (float)((double)d0 * (double)d0);
    float gd = (float)((double)d1 * (double)d1);
    float bd = (float)((double)d2 * (double)d2);
    return n_sqrtf(rd + gd + bd);
}

// -------------------------------------------------------------------------------------------
// geometric distance + convexity (src/clustering.cpp:53-96)
// -------------------------------------------------------------------------------------------
F3DS_HD void n_unit_c(const float c1[3], const float c2[3], float C[3]) {
    C[0] = c1[0] - c2[0]; C[1] = c1[1] - c2[1]; C[2] = c1[2] - c2[2];
    float n = n_sqrtf(n_sum3(C[0] * C[0], C[1] * C[1], C[2] * C[2]));
    C[0] /= n; C[1] /= n; C[2] /= n;
}
F3DS_HD float n_normals_diff(const float n1[3], const float c1[3], const float n2[3], const float c2[3]) {
    float C[3]; n_unit_c(c1, c2, C);
    float cr[3]; n_cross(n1[0], n1[1], n1[2], n2[0], n2[1], n2[2], cr);
    float N1xN2 = n_sqrtf(n_sum3(cr[0] * cr[0], cr[1] * cr[1], cr[2] * cr[2]));
    float N1_C = m_absf(n_sum3(n1[0] * C[0], n1[1] * C[1], n1[2] * C[2]));
    float N2_C = m_absf(n_sum3(n2[0] * C[0], n2[1] * C[1], n2[2] * C[2]));
    return (N1xN2 + N1_C + N2_C) / 3;
}
F3DS_HD bool n_is_convex(const float n1[3], const float c1[3], const float n2[3], const float c2[3]) {
    float C[3]; n_unit_c(c1, c2, C);
    float cos1 = n_sum3(n1[0] * C[0], n1[1] * C[1], n1[2] * C[2]);
    float cos2 = n_sum3(n2[0] * C[0], n2[1] * C[1], n2[2] * C[2]);
    return cos1 >= cos2;
}

// region record used by the merge stage: 16 floats
//   [0..2] centroid  [3..5] normal  [6..8] mean rgb  [9..11] Lab of the mean  [12..15] spare
// delta_c_g (src/clustering.cpp:107-142): first = colour, second = geometry
template <class K = m_lit> F3DS_HD void n_delta_c_g(const float* r1, const float* r2, int color_metric, int geom_metric, float* dc, float* dg, K mc = K()) {
    float c;
    if (color_metric == 0) c = n_ciede00(r1 + 9, r2 + 9, mc) / F3DS_LAB_RANGE;
    else c = n_rgb_eucl(r1 + 6, r2 + 6) / F3DS_RGB_RANGE;
    float g = n_normals_diff(r1 + 3, r1, r2 + 3, r2);
    if (geom_metric == 1 && n_is_convex(r1 + 3, r1, r2 + 3, r2)) g *= 0.5;
    *dc = c; *dg = g;
}

// multimap<float,...> order with NaN after every number (the reference's behaviour with NaN
// keys is undefined); -0 and +0 compare equal like operator<
F3DS_HD uint32_t n_weight_key(float w) {
    if (w != w) return 0xFFFFFFFEu;      // 0xFFFFFFFF is left free for "no edge"
    uint32_t b = m_bitsf(w);
    if (b == 0x80000000u) b = 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

}  // namespace f3ds
#endif  // F3DS_NUMERICS_H_
