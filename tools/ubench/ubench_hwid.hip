// Which SIMD does wave k of a workgroup run on?  Prints HW_ID fields of every wave of a few workgroups (gfx9 layout: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13).
// build: hipcc --offload-arch=gfx950 -O2 -o ubench_hwid ubench_hwid.hip ; run: ./ubench_hwid [threads per block] [blocks] [dynamic LDS bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned* out) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = id; out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = xcc; }
}
int main(int argc, char** argv) {
    int T = argc > 1 ? atoi(argv[1]) : 512, B = argc > 2 ? atoi(argv[2]) : 4, lds = argc > 3 ? atoi(argv[3]) : 1024;
    unsigned* d; hipMalloc(&d, B * (T / 64) * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(B), dim3(T), lds, 0, d);
    unsigned* h = (unsigned*)malloc(B * (T / 64) * 8); hipMemcpy(h, d, B * (T / 64) * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < B; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < T / 64; ++w) { unsigned id = h[(b * (T / 64) + w) * 2]; printf("  w%d simd %u slot %u cu %u se %u xcc %u |", w, (id >> 4) & 3, id & 15, (id >> 8) & 15, (id >> 13) & 7, h[(b * (T / 64) + w) * 2 + 1] & 15); }
        printf("\n");
    }
    return 0;
}
