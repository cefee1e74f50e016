// ubench_fetch.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this path
// (guides/MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access
// widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Kernels with a known number of bytes moved, each over a working set far beyond the 256 MB Infinity Cache:
//   k_stream16   every lane reads 16 consecutive bytes (global_load_dwordx4)            bytes = N * 16
//   k_stream4    every lane reads 4 consecutive bytes                                   bytes = N * 4
//   k_gather4    every lane reads 4 bytes at a pseudo-random 64-B aligned offset        lines touched = N (distinct w.h.p.)
//   k_gather16   every lane reads 16 bytes at a pseudo-random 64-B aligned offset       lines touched = N
//   k_write4 / k_write16  every lane writes 4 / 16 consecutive bytes                    bytes = N * 4 / 16
//   k_scatter4   every lane writes 4 bytes at a pseudo-random 64-B aligned offset       lines touched = N
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); tools/pmc_calibrate.py prints bytes per counter unit.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void k_stream16(const uint4* a, uint32_t* sink, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint4 v = a[i]; if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = 1; } }
__global__ void k_stream4(const uint32_t* a, uint32_t* sink, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { if (a[i] == 0x12345678u) sink[0] = 1; } }
__global__ void k_gather4(const uint32_t* a, uint32_t* sink, size_t n, size_t lines) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { if (a[(mix(i) % lines) * 16] == 0x12345678u) sink[0] = 1; } }
__global__ void k_gather16(const uint4* a, uint32_t* sink, size_t n, size_t lines) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { uint4 v = a[(mix(i) % lines) * 4]; if ((v.x ^ v.w) == 0x12345678u) sink[0] = 1; } }
__global__ void k_write4(uint32_t* a, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) a[i] = (uint32_t)i; }
__global__ void k_write16(uint4* a, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) a[i] = make_uint4((uint32_t)i, 1, 2, 3); }
__global__ void k_scatter4(uint32_t* a, size_t n, size_t lines) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) a[(mix(i) % lines) * 16] = (uint32_t)i; }
int main() {
    const size_t bytes = (size_t)4 << 30;           // 4 GiB working set
    void* a; uint32_t* sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0, bytes);
    const size_t lines = bytes / 64;
    const size_t n16 = bytes / 16, n4 = bytes / 4 / 4 /* 1 GiB of 4-byte reads */, ng = (size_t)32 << 20;
    const int B = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_stream16, dim3((n16 + B - 1) / B), dim3(B), 0, 0, (const uint4*)a, sink, n16);
        hipLaunchKernelGGL(k_stream4, dim3((n4 + B - 1) / B), dim3(B), 0, 0, (const uint32_t*)a, sink, n4);
        hipLaunchKernelGGL(k_gather4, dim3((ng + B - 1) / B), dim3(B), 0, 0, (const uint32_t*)a, sink, ng, lines);
        hipLaunchKernelGGL(k_gather16, dim3((ng + B - 1) / B), dim3(B), 0, 0, (const uint4*)a, sink, ng, lines);
        hipLaunchKernelGGL(k_write16, dim3((n16 + B - 1) / B), dim3(B), 0, 0, (uint4*)a, n16);
        hipLaunchKernelGGL(k_write4, dim3((n4 + B - 1) / B), dim3(B), 0, 0, (uint32_t*)a, n4);
        hipLaunchKernelGGL(k_scatter4, dim3((ng + B - 1) / B), dim3(B), 0, 0, (uint32_t*)a, ng, lines);
        hipDeviceSynchronize();
    }
    printf("expected: k_stream16 %zu B, k_stream4 %zu B, k_gather4 / k_gather16 %zu lines of 64 B, k_write16 %zu B, k_write4 %zu B, k_scatter4 %zu lines\n", n16 * 16, n4 * 4, ng, n16 * 16, n4 * 4, ng);
    return 0;
}
