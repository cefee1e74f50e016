// Does a dependent chain that walks across lanes with DPP (row_ror:1 on the accumulator) need wait states between a VALU write and the DPP read of the same
// register on gfx950, and what does it cost?  Running colour mean acc' = acc + inv * (c - acc) over n rows, three ways: plain (one lane per channel), DPP walk without
// and with s_nop 1.  Prints cycles per row and whether the results agree bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k_plain(const float* c, const float* inv, float* out, unsigned long long* cyc, int n) {
    float acc = 0.25f;
    const int lane = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < n; ++j) { const float cj = c[j * 4 + (lane & 3)], ij = inv[j]; acc = acc + ij * (cj - acc); }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane < 3) out[lane] = acc;
    if (lane == 0) cyc[0] = t1 - t0;
}
template <int NOP>
__global__ void k_walk(const float* c, const float* inv, float* out, unsigned long long* cyc, int n) {      // n a multiple of 16
    const int lane = threadIdx.x, row = lane >> 4, l = lane & 15;
    float acc = 0.25f;      // valid in lane 15 of every row before the first step
    unsigned long long total = 0;
    for (int j0 = 0; j0 < n; j0 += 16) {
        const float D = c[(j0 + l) * 4 + (row & 3)], I = inv[j0 + l];
        float t;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (NOP == 0) asm volatile("v_subrev_f32_dpp %1, %0, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mul_f32 %1, %3, %1\n\tv_add_f32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(acc), "=&v"(t) : "v"(D), "v"(I));
            else asm volatile("s_nop 1\n\tv_subrev_f32_dpp %1, %0, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mul_f32 %1, %3, %1\n\tv_add_f32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(acc), "=&v"(t) : "v"(D), "v"(I));
        }
        total += __builtin_amdgcn_s_memtime() - t0;
    }
    if (l == 15 && row < 3) out[row] = acc;
    if (lane == 0) cyc[0] = total;
}
int main() {
    const int n = 16000;
    std::vector<float> hc(n * 4), hi(n);
    unsigned s = 12345;
    for (int j = 0; j < n; ++j) { for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; hc[j * 4 + k] = (float)(s >> 8) / 65536.0f; } hi[j] = 1.0f / (float)(j + 2); }
    float *dc, *di, *dout; unsigned long long* dcy;
    (void)hipMalloc(&dc, n * 16); (void)hipMalloc(&di, n * 4); (void)hipMalloc(&dout, 64); (void)hipMalloc(&dcy, 8);
    (void)hipMemcpy(dc, hc.data(), n * 16, hipMemcpyHostToDevice); (void)hipMemcpy(di, hi.data(), n * 4, hipMemcpyHostToDevice);
    float r[3][3]; unsigned long long cy[3];
    k_plain<<<1, 64>>>(dc, di, dout, dcy, n); (void)hipMemcpy(r[0], dout, 12, hipMemcpyDeviceToHost); (void)hipMemcpy(&cy[0], dcy, 8, hipMemcpyDeviceToHost);
    k_walk<0><<<1, 64>>>(dc, di, dout, dcy, n); (void)hipMemcpy(r[1], dout, 12, hipMemcpyDeviceToHost); (void)hipMemcpy(&cy[1], dcy, 8, hipMemcpyDeviceToHost);
    k_walk<1><<<1, 64>>>(dc, di, dout, dcy, n); (void)hipMemcpy(r[2], dout, 12, hipMemcpyDeviceToHost); (void)hipMemcpy(&cy[2], dcy, 8, hipMemcpyDeviceToHost);
    const char* names[3] = {"plain (loads in the loop)", "DPP walk, no wait states", "DPP walk, s_nop 1 per row"};
    for (int v = 0; v < 3; ++v) printf("%-28s %.2f cycles per row   result %.9g %.9g %.9g   %s\n", names[v], (double)cy[v] / n, r[v][0], r[v][1], r[v][2], memcmp(r[v], r[0], 12) ? "DIFFERS from plain" : "same bits as plain");
    return 0;
}
