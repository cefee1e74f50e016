// ubench_math.hip -- latency of the building blocks of the merge loop on one wave (MI355X): dependent chains of basic f64 / f32
// operations, the transcendental functions of csrc/f3ds_math.h, the colour distance, the plane normal, rgb2lab, and
// LDS / global round trips.  Shader clocks (s_memtime) per dependent step.  Development tool: nothing links against it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../fast-3d-pointcloud-segmentation_amd/csrc/f3ds_algo.h"
using namespace f3ds;

#define N 64
__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

#include "../../fast-3d-pointcloud-segmentation_amd/csrc/f3ds_quad.h"

__global__ void k(double* out, unsigned long long* t, const uint32_t* chase, double seed, int active_waves) {
    __shared__ uint32_t lds[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (i * 37 + 11) & 1023;
    __syncthreads();
    // active_waves > 0: only that many waves run the chains, the others of the block wait at a barrier (what most waves of a merge workgroup do most of the time)
    if (active_waves > 0 && (int)(threadIdx.x >> 6) >= active_waves) { __syncthreads(); return; }
    double x = seed + lane * 1e-3; float xf = (float)x;
    uint32_t p = lane;
    int r = 0;
    unsigned long long t0;
// the asm statements tie the clock reads to the chain's input and result (the scheduler would otherwise move the arithmetic past them)
#define RUN(name, expr) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x), "+v"(xf), "+v"(p) :: "memory"); t0 = now(); asm volatile("" : "+v"(x), "+v"(xf), "+v"(p) : "s"(t0) : "memory"); \
    for (int i = 0; i < N; ++i) { expr; } asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x), "+v"(xf), "+v"(p) :: "memory"); t[r++] = now() - t0; }
    RUN("f64 add", x = x + 1.0000001)
    RUN("f64 mul", x = x * 1.0000001)
    RUN("f64 mul+add (no fma)", x = x * 1.0000001 + 0.5)
    RUN("f64 fma", x = __builtin_fma(x, 1.0000001, 0.5))
    x = seed;
    RUN("f64 div", x = 3.0 / x + 1.0)
    RUN("f64 sqrt", x = n_sqrt(x + 2.0))
    RUN("f32 add", xf = xf + 1.0001f)
    RUN("f32 div", xf = 3.0f / xf + 1.0f)
    RUN("f32 sqrt", xf = n_sqrtf(xf + 2.0f))
    RUN("m_exp", x = m_exp(x * 0.1) )
    RUN("m_log", x = m_log(x + 2.0))
    RUN("m_sin", x = m_sin(x + 1.0))
    RUN("m_cos", x = m_cos(x + 1.0))
    RUN("m_atan2", x = m_atan2(x + 0.3, 1.7))
    RUN("m_pow_pos(x,2.4)", x = m_pow_pos(x + 1.1, 2.4) * 0.1)
    RUN("m_cbrt_pos", x = m_cbrt_pos(x + 1.1))
    {
        float l1[3] = {50.0f + (float)x, 2.5f, -10.0f}, l2[3] = {60.0f, -3.0f, 20.0f};
        RUN("n_ciede00 (one lane)", (l1[1] = 2.0f + n_ciede00(l1, l2) * 0.1f, l2[2] = 20.0f - l1[1]))
        x += l1[0];
    }
    {
        float l1[3] = {50.0f + (float)x, 2.5f, -10.0f}, l2[3] = {60.0f, -3.0f, 20.0f};
        RUN("n_ciede00_quad", (l1[1] = 2.0f + n_ciede00_quad(l1, l2, lane & 3) * 0.1f, l2[2] = 20.0f - l1[1]))
        x += l1[0];
    }
    {
        float acc[9] = {1.1f, 0.2f, 0.3f, 2.2f, 0.1f, 3.3f, 0.5f, 0.6f, 0.7f}; float cen[3] = {0.1f, 0.2f, 1.0f}, n4[4];
        RUN("n_plane_normal", (n_plane_normal(acc, 50u, cen, n4), acc[0] = 1.1f + n4[0] * 0.01f, acc[4] = 0.1f + n4[1] * 0.01f))
        x += acc[0];
    }
    {
        float rgb[3] = {120.0f, 60.0f, 200.0f}, lab[3];
        RUN("n_rgb2lab", (n_rgb2lab(rgb, lab), rgb[0] = 100.0f + lab[1] * 0.1f, rgb[1] = 60.0f + lab[2] * 0.1f, rgb[2] = 150.0f + lab[0] * 0.1f))
        x += rgb[0];
    }
    {
        float ch = 120.0f - 30.0f * (lane % 3), lab[3];
        RUN("rgb -> Lab on three lanes", (lab_three_lanes(ch, lane, lab), ch = 100.0f + lab[1] * 0.1f + (lane % 3)))
        x += ch;
    }
    {
        float r1[16] = {0.1f, 0.2f, 1.0f, 0.0f, 0.6f, 0.8f, 100, 50, 20, 50.0f, 2.5f, -10.0f}, r2[16] = {0.3f, 0.1f, 1.2f, 0.6f, 0.0f, 0.8f, 90, 40, 30, 60.0f, -3.0f, 20.0f};
        RUN("n_normals_diff + is_convex", (r1[0] = 0.1f + n_normals_diff(r1 + 3, r1, r2 + 3, r2) + (n_is_convex(r1 + 3, r1, r2 + 3, r2) ? 0.01f : 0.0f)))
        x += r1[0];
    }
    RUN("LDS dependent read", p = lds[p])
    RUN("LDS atomicAdd (distinct addresses) dependent", p = atomicAdd(&lds[p & 1023], 1u) & 1023)
    RUN("LDS atomicAdd (one address, 64 lanes)", p = atomicAdd(&lds[0], 1u) & 1023)
    RUN("global dependent load (L2 hit)", p = chase[p])
    RUN("quad_bcast f64 x2", x = quad_bcast<0>(x) + quad_bcast<1>(x))
    if (active_waves == 0) RUN("s_barrier (this block)", __syncthreads())
    else if (threadIdx.x < 64) { t[r++] = 0; }
    out[threadIdx.x] = x + xf + p;
    t[63] = r;
    if (active_waves > 0) __syncthreads();
}

int main() {
    const char* names[] = {"f64 add", "f64 mul", "f64 mul+add (no fma)", "f64 fma", "f64 div", "f64 sqrt", "f32 add", "f32 div", "f32 sqrt", "m_exp", "m_log", "m_sin", "m_cos", "m_atan2",
                           "m_pow_pos(x,2.4)", "m_cbrt_pos", "n_ciede00 (one lane)", "n_ciede00_quad", "n_plane_normal", "n_rgb2lab", "rgb -> Lab on three lanes", "n_normals_diff + is_convex", "LDS dependent read",
                           "LDS atomicAdd (distinct addresses)", "LDS atomicAdd (one address, 64 lanes)", "global dependent load (L2 hit)", "quad_bcast f64 x2", "s_barrier"};
    double* out; unsigned long long* t; uint32_t* chase;
    hipMalloc(&out, 8 * 1024); hipMalloc(&t, 8 * 64); hipMalloc(&chase, 4 * 65536);
    uint32_t h[65536]; for (int i = 0; i < 65536; ++i) h[i] = (uint32_t)((i * 40503u + 977u) & 65535u);
    hipMemcpy(chase, h, sizeof h, hipMemcpyHostToDevice);
    const int cfg[4][2] = {{64, 0}, {512, 0}, {512, 1}, {512, 4}};
    for (int c = 0; c < 4; ++c) {
        const int threads = cfg[c][0], active = cfg[c][1];
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, t, chase, 1.37, active); hipDeviceSynchronize(); }
        unsigned long long ht[64]; hipMemcpy(ht, t, sizeof ht, hipMemcpyDeviceToHost);
        printf("---- %d threads in the block, %s, shader clocks per dependent step (wave 0) ----\n", threads, active == 0 ? "all waves run the chains" : (active == 1 ? "wave 0 runs the chains, the others wait at a barrier" : "waves 0-3 run the chains, the others wait at a barrier"));
        for (int i = 0; i < (int)ht[63] && i < 28; ++i) printf("%-48s %8.1f\n", names[i], (double)ht[i] / N);
    }
    return 0;
}
