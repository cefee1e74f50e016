#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN(NAME, BODY) \
__global__ void NAME(float* out, unsigned long long* cyc, int n) { \
    float acc = out[threadIdx.x]; float c = out[64 + threadIdx.x], one = out[128 + threadIdx.x]; double dacc = acc, dc = c; (void)dacc; (void)dc; (void)one; \
    unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int r = 0; r < n; ++r) { BODY BODY BODY BODY BODY BODY BODY BODY } \
    unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
    out[threadIdx.x] = acc + (float)dacc; if (threadIdx.x == 0) cyc[0] = t1 - t0; }
CHAIN(k_add,  asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(c));)
CHAIN(k_sub,  asm volatile("v_sub_f32 %0, %1, %0" : "+v"(acc) : "v"(c));)
CHAIN(k_mul,  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc) : "v"(one));)
CHAIN(k_fma,  asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(acc) : "v"(c), "v"(one));)
CHAIN(k_fmac, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(c), "v"(one));)
CHAIN(k_add64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(dacc) : "v"(dc));)
CHAIN(k_mov,  asm volatile("v_mov_b32 %0, %0" : "+v"(acc));)
CHAIN(k_addu, asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(c));)
CHAIN(k_add_e64, asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(acc) : "v"(c));)
CHAIN(k_add_dpp, asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c));)
CHAIN(k_add_nop, asm volatile("v_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(acc) : "v"(c));)
CHAIN(k_add_2indep, asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1" : "+v"(acc), "+v"(one) : "v"(c));)
int main() {
    float* d; unsigned long long* c; (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 64); (void)hipMemset(d, 0, 4096);
    unsigned long long h;
#define RUN(K, ops) for (int rep = 0; rep < 2; ++rep) { K<<<1, 64>>>(d, c, 10000); (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf(#K ": %.2f cycles per step (%d instr)\n", (double)h / 80000.0, ops); }
    RUN(k_add, 1) RUN(k_sub, 1) RUN(k_mul, 1) RUN(k_fma, 1) RUN(k_fmac, 1) RUN(k_add64, 1) RUN(k_mov, 1) RUN(k_addu, 1) RUN(k_add_e64, 1) RUN(k_add_dpp, 1) RUN(k_add_nop, 2) RUN(k_add_2indep, 2)
    return 0;
}
