// ubench_consts.hip -- what do the branches and the f64 literals of csrc/f3ds_math.h cost a lone wave?  m_exp three ways: as it was until the middle of round 2 (early
// returns, literals), branch-free (selects) with literals, branch-free with the coefficients read from an LDS table.  All three return the same bits (checked here on a
// sweep).  Then the header's functions as they are now, with literals (m_lit; a loop hoists them, the merge loop cannot) and from the LDS table (m_tab).
// Shader clocks per dependent step on one wave.  Development tool: nothing links against it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include "../../fast-3d-pointcloud-segmentation_amd/csrc/f3ds_math.h"
using namespace f3ds;

#define N 64
__host__ __device__ inline double exp_r2(double x) {      // round 2's m_exp
    if (!(x <= 709.782712893384)) return x != x ? x : m_inf();
    if (x < -745.2) return 0.0;
    const double INV_LN2 = 1.4426950408889634, LN2_HI = 0x1.62e42fefa39efp-1, LN2_LO = 0x1.abc9e3b39803fp-56;
    const double MAGIC = 0x1.8p52;
    const double s = x * INV_LN2 + MAGIC;
    const int k = (int)(uint32_t)m_bits(s);
    const double kd = s - MAGIC;
    const double r = m_fma(-kd, LN2_LO, m_fma(-kd, LN2_HI, x));
    const double p = m_estrin14(1.0, 1.0, 0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0, 1.0 / 40320.0, 1.0 / 362880.0,
                                1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0, r);
    if (k > 1000) return (p * m_pow2(1000)) * m_pow2(k - 1000);
    if (k < -1000) return (p * m_pow2(-1000)) * m_pow2(k + 1000);
    return p * m_pow2(k);
}
__host__ __device__ inline double pow2u(int k) { return m_from_bits((uint64_t)((uint32_t)k + 1023u) << 52); }
struct Lit { __host__ __device__ double operator[](int i) const {
    constexpr double t[20] = {1.0, 1.0, 0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0, 1.0 / 40320.0, 1.0 / 362880.0,
                              1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0, 1.4426950408889634, 0x1.8p52, 0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56, 0, 0};
    return t[i]; } };
struct Tab { const double* p; __device__ double operator[](int i) const { return p[i]; } };
template <class C> __host__ __device__ inline double exp_bf(double x, C c) {      // branch-free
    const double MAGIC = c[15];
    const double s = x * c[14] + MAGIC;
    const int k = (int)(uint32_t)m_bits(s);
    const double kd = s - MAGIC;
    const double r = m_fma(-kd, c[17], m_fma(-kd, c[16], x));
    const double p = m_estrin14(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11], c[12], c[13], r);
    const int k1 = k > 1000 ? 1000 : (k < -1000 ? -1000 : k);
    double v = (p * pow2u(k1)) * pow2u(k - k1);
    v = x < -745.2 ? 0.0 : v;
    v = !(x <= 709.782712893384) ? (x != x ? x : m_inf()) : v;
    return v;
}

__global__ void k(double* out, unsigned long long* t, double seed) {
    __shared__ double tab[20];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 20) tab[threadIdx.x] = Lit()[threadIdx.x];
    __syncthreads();
    double x = seed + lane * 1e-3;
    int r = 0;
    unsigned long long t0;
#define RUN(expr) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x) :: "memory"); t0 = __builtin_amdgcn_s_memtime(); asm volatile("" : "+v"(x) : "s"(t0) : "memory"); \
    for (int i = 0; i < N; ++i) { expr; } asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x) :: "memory"); t[r++] = __builtin_amdgcn_s_memtime() - t0; }
    RUN(x = exp_r2(x * 0.1))
    RUN(x = exp_bf(x * 0.1, Lit()))
    RUN(x = exp_bf(x * 0.1, Tab{tab}))
    RUN(x = m_exp(x * 0.1))
    RUN(x = m_log(x + 2.0))
    RUN(x = m_pow_pos(x + 1.1, 2.4) * 0.1)
    RUN(x = m_cbrt_pos(x + 1.1))
    RUN(x = m_sin(x + 1.0))
    RUN(x = m_cos(x + 1.0))
    RUN(x = m_atan2(x + 0.3, 1.7))
    __shared__ double mtab[MC_COUNT];
    m_table_fill(mtab, threadIdx.x, 64);
    __syncthreads();
    const m_tab mc{mtab};
    RUN(x = m_exp(x * 0.1, mc))
    RUN(x = m_log(x + 2.0, mc))
    RUN(x = m_pow_pos(x + 1.1, 2.4, mc) * 0.1)
    RUN(x = m_cbrt_pos(x + 1.1, mc))
    RUN(x = m_sin(x + 1.0, mc))
    RUN(x = m_atan2(x + 0.3, 1.7, mc))
    out[threadIdx.x] = x;
    t[63] = r;
}
__global__ void check(const double* in, double* o1, double* o2, double* o3, int n) {
    __shared__ double tab[20];
    if (threadIdx.x < 20) tab[threadIdx.x] = Lit()[threadIdx.x];
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { o1[i] = exp_r2(in[i]); o2[i] = exp_bf(in[i], Lit()); o3[i] = exp_bf(in[i], Tab{tab}); }
}

int main() {
    const char* names[] = {"m_exp round 2 (early returns, literals)", "m_exp branch-free, literals", "m_exp branch-free, coefficients from LDS", "m_exp (header)", "m_log (header)", "m_pow_pos (header)", "m_cbrt_pos (header)", "m_sin (header)", "m_cos (header)", "m_atan2 (header)",
                           "m_exp (header, LDS table)", "m_log (header, LDS table)", "m_pow_pos (header, LDS table)", "m_cbrt_pos (header, LDS table)", "m_sin (header, LDS table)", "m_atan2 (header, LDS table)"};
    double* out; unsigned long long* t;
    hipMalloc(&out, 8 * 1024); hipMalloc(&t, 8 * 64);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, 1.37); hipDeviceSynchronize(); }
    unsigned long long ht[64]; hipMemcpy(ht, t, sizeof ht, hipMemcpyDeviceToHost);
    for (int i = 0; i < (int)ht[63]; ++i) printf("%-48s %8.1f\n", names[i], (double)ht[i] / N);
    const int n = 1 << 20;
    double* h = new double[n]; double *d, *o1, *o2, *o3;
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const double u = (double)(s >> 11) / 9007199254740992.0;
        h[i] = i < 16 ? (double[]){0.0, -0.0, 709.78, 709.79, -745.1, -745.3, -744.0, 700.0, 1e308, -1e308, NAN, INFINITY, -INFINITY, 1e-310, -708.5, 709.782712893384}[i] : (i & 1 ? -760.0 + u * 1480.0 : -3.0 + u * 6.0); }
    hipMalloc(&d, 8 * n); hipMalloc(&o1, 8 * n); hipMalloc(&o2, 8 * n); hipMalloc(&o3, 8 * n);
    hipMemcpy(d, h, 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(256), dim3(256), 0, 0, d, o1, o2, o3, n); hipDeviceSynchronize();
    double *r1 = new double[n], *r2 = new double[n], *r3 = new double[n];
    hipMemcpy(r1, o1, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(r3, o3, 8 * n, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) { const double e = exp_r2(h[i]), e2 = exp_bf(h[i], Lit());
        if (memcmp(&r1[i], &e, 8) || memcmp(&r2[i], &e, 8) || memcmp(&r3[i], &e, 8) || memcmp(&e2, &e, 8)) { if (bad++ < 5) printf("mismatch at x = %a: %a %a %a host %a %a\n", h[i], r1[i], r2[i], r3[i], e, e2); } }
    printf("bit comparison of the three forms on %d arguments (device and host): %d mismatches\n", n, bad);
    return bad != 0;
}
