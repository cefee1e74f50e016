// Micro-benchmark: what a vector-memory instruction costs the texture-addresser / L1 as a function of active lanes and distinct lines.
// Every wave issues ITER independent 4-byte loads (or stores) per lane pattern from an L2-resident table; all CUs busy, 8 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o ta_cost ta_cost.hip ; run: ./ta_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 256;
// mode: 0 = every lane its own line (random), 1 = 64 consecutive words (one coalesced 256 B), 2 = every lane its own line but only `active` lanes on,
// 3 = 16-byte loads, every lane its own line, 4 = 4 lanes per line (16 lines per instruction)
template <int MODE, bool STORE>
__global__ void k(uint32_t* tbl, uint32_t mask_words, int active, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0, x = wave * 2654435761u + 12345u;
    if (MODE == 2 && (int)lane >= active) return;
    for (int i = 0; i < ITER; ++i) {
        x = x * 1664525u + 1013904223u;
        uint32_t idx;
        if (MODE == 0 || MODE == 2 || MODE == 3) idx = ((x >> 4) + lane * 7919u * 32u) & mask_words;      // a different line per lane
        else if (MODE == 1) idx = ((x >> 4) & mask_words & ~63u) + lane;                                  // one 256-byte run
        else idx = (((x >> 4) + (lane >> 2) * 7919u * 32u) & mask_words & ~3u) + (lane & 3u);            // four lanes per 16 bytes
        if (MODE == 3) idx &= ~3u;
        if (STORE) tbl[idx] = x;
        else if (MODE == 3) { const uint4 v = *reinterpret_cast<const uint4*>(tbl + idx); acc += v.x + v.w; }
        else acc += tbl[idx];
    }
    if (!STORE && acc == 0x12345678u) out[0] = acc;
}
template <int MODE, bool STORE>
int run(const char* name, uint32_t* tbl, uint32_t words, int active, uint32_t* out) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 256 * 8, threads = 256;      // 8 workgroups of 4 waves per CU
    k<MODE, STORE><<<blocks, threads>>>(tbl, words - 1, active, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) k<MODE, STORE><<<blocks, threads>>>(tbl, words - 1, active, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double instr = (double)blocks * (threads / 64) * ITER;          // wave-instructions
    const double per_cu_ns = ms * 1e6 / (instr / 256.0);
    printf("%-58s table %5.1f MB: %7.3f ms  %6.1f ns per wave-instruction per CU (= %5.0f cycles at 2.4 GHz)  %6.1f G lane-ops/s\n", name, words * 4 / 1048576.0, ms, per_cu_ns, per_cu_ns * 2.4,
           instr * (MODE == 2 ? active : 64) / ms / 1e6);
    return 0;
}
int main() {
    uint32_t *tbl, *out;
    const uint32_t big = 64u << 20;      // words
    CK(hipMalloc(&tbl, (size_t)big * 4)); CK(hipMalloc(&out, 64)); CK(hipMemset(tbl, 1, (size_t)big * 4));
    for (uint32_t words : {256u << 10, 4u << 20}) {      // 1 MB (L2-resident), 16 MB (spills one XCD's L2)
        run<1, false>("load  dword, 64 lanes, one 256-byte run", tbl, words, 64, out);
        run<4, false>("load  dword, 64 lanes, 16 lines (4 lanes each)", tbl, words, 64, out);
        run<0, false>("load  dword, 64 lanes, 64 lines", tbl, words, 64, out);
        run<2, false>("load  dword, 32 lanes active, 32 lines", tbl, words, 32, out);
        run<2, false>("load  dword, 16 lanes active, 16 lines", tbl, words, 16, out);
        run<2, false>("load  dword,  4 lanes active,  4 lines", tbl, words, 4, out);
        run<3, false>("load  dwordx4, 64 lanes, 64 lines", tbl, words, 64, out);
        run<1, true>("store dword, 64 lanes, one 256-byte run", tbl, words, 64, out);
        run<0, true>("store dword, 64 lanes, 64 lines", tbl, words, 64, out);
        run<2, true>("store dword, 16 lanes active, 16 lines", tbl, words, 16, out);
    }
    return 0;
}
