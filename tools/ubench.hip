// ubench.hip — dependent-chain latency of the VALU ops the coder's serial step is made of,
// for ONE wave alone on a SIMD (the coder's regime: 239 waves on 1024 SIMDs).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 2048
#define CHAIN(NAME, BODY)                                                                    \
    __global__ void NAME(uint32_t *out, uint64_t *cyc, uint32_t a, uint32_t b) {             \
        uint32_t x = a + threadIdx.x, y = b;                                                 \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                          \
        _Pragma("unroll 16") for (int i = 0; i < REP; i++) { BODY; }                         \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                          \
        out[threadIdx.x] = x + y;                                                            \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                              \
    }

CHAIN(k_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y)))
CHAIN(k_mulhi, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(y)))
CHAIN(k_mullo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y)))
CHAIN(k_mul24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(y)))
CHAIN(k_mulhi24, asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(y)))
CHAIN(k_ffbh, asm volatile("v_ffbh_u32 %0, %0" : "+v"(x)))
CHAIN(k_bfi, asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x) : "v"(y)))
CHAIN(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc"))
CHAIN(k_2indep, asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(x), "+v"(y) : "v"(a)))
CHAIN(k_add_salu, asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(x) : "v"(y) : "s20"))
CHAIN(k_lshl64, uint64_t q = ((uint64_t)y << 32) | x; asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q)); x = (uint32_t)q; y = (uint32_t)(q >> 32))

// VALU throughput: every thread runs 8 independent v_add chains; waves per SIMD = threads/256.
__global__ void k_tp(uint32_t *out, uint32_t a, int iters) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                     "v_xor_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_lshlrev_b32 %6, 1, %6\n v_sub_u32 %7, %7, %8"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    }
    if ((x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7) == 0x12345u) out[0] = 1;
}

int main() {
    uint32_t *out; uint64_t *cyc;
    hipMalloc(&out, 256); hipMalloc(&cyc, 8);
    struct { const char *n; void (*k)(uint32_t *, uint64_t *, uint32_t, uint32_t); int ops; } T[] = {
        {"v_add_u32 (dependent)", k_add, 1}, {"v_mul_hi_u32", k_mulhi, 1}, {"v_mul_lo_u32", k_mullo, 1}, {"v_mul_u32_u24", k_mul24, 1},
        {"v_mul_hi_u32_u24", k_mulhi24, 1}, {"v_ffbh_u32", k_ffbh, 1}, {"v_bfi_b32", k_bfi, 1}, {"v_cndmask_b32", k_cndmask, 1},
        {"2 independent v_add chains (per pair)", k_2indep, 1}, {"v_add + s_add (per pair)", k_add_salu, 1}, {"v_lshlrev_b64", k_lshl64, 1}};
    for (auto &t : T) {
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(t.k, dim3(1), dim3(64), 0, 0, out, cyc, 3u, 5u);
        uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-42s %7.2f s_memtime ticks per iteration\n", t.n, (double)c / REP);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;
    for (int threads : {64, 256, 512, 1024}) {
        for (int wg : {1, 256, 512}) {
            if (threads == 64 && wg != 256) continue;
            hipLaunchKernelGGL(k_tp, dim3(wg), dim3(threads), 0, 0, out, 3u, 1000);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_tp, dim3(wg), dim3(threads), 0, 0, out, 3u, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double waves = (double)wg * threads / 64.0;
            printf("k_tp wg=%4d threads=%4d: %.3f ms, %.2f ns per wave-instruction per wave, chip %.3e wave-instr/s\n", wg, threads, ms,
                   ms * 1e6 / (iters * 8.0), waves * iters * 8.0 / (ms * 1e-3));
        }
    }
    return 0;
}
