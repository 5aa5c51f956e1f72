// ubench_mem3.hip — cache-policy bits on the rank kernels' store pattern: 16-byte stores scattered over a block's 1 MiB stream
// region, 64 slices of a block written by 64 different wavefronts at about the same time, blocks handed out in order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__device__ __forceinline__ void st16(uint4 *p, uint4 vv) {
    const u32x4 v = {vv.x, vv.y, vv.z, vv.w};
    if (MODE == 0) *p = vv;
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
}

// job = (block, slice): 1024 positions of the block, positions = odd-multiplier permutation of 0..65535 (distinct over the block)
template <int MODE, int PAIR>
__global__ void __launch_bounds__(64) k_scatter(uint4 *P, uint32_t nblocks, uint32_t *counter) {
    const uint32_t lane = threadIdx.x;
    for (;;) {
        uint32_t job = 0;
        if (lane == 0) job = atomicAdd(counter, 1u);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= nblocks * 64u) break;
        const uint32_t b = job >> 6, sl = job & 63u;
        uint4 *dst = P + (size_t)b * 65536u;
        for (uint32_t r = 0; r < 16; r++) {
            const uint32_t k = sl * 1024u + r * 64u + lane;
            uint32_t pos = (k * 40503u) & 0xFFFFu;
            if (PAIR == 1) pos = ((((k >> 1) * 40503u) & 0x7FFFu) << 1) | (k & 1u);   // pairs of lanes hit one 32-byte sector
            if (PAIR == 2) pos = (k * 6u + ((k * 2654435761u) >> 30)) & 0xFFFFu;          // a big group: ascending positions ~6 apart (lanes share 128-byte lines)
            if (PAIR == 3) {   // the same records, but one store instruction takes lane l's record from round (r + l) & 7 of its batch of 8 rounds
                const uint32_t rr = (r & 8u) | ((r + lane) & 7u);
                const uint32_t k2 = sl * 1024u + rr * 64u + lane;
                pos = (k2 * 6u + ((k2 * 2654435761u) >> 30)) & 0xFFFFu;
            }
            if (PAIR == 4) pos = (k * 37u + ((k * 2654435761u) >> 30)) & 0xFFFFu;         // a medium group: ~37 apart (every lane its own 128-byte line, nearby)
            st16<MODE>(dst + pos, make_uint4(k, b, sl, r));
        }
    }
}

template <int MODE, int PAIR>
int run(uint4 *P, uint32_t nb, uint32_t *cnt, const char *name) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 3; it++) {
        CHECK(hipMemsetAsync(cnt, 0, 4, 0));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_scatter<MODE, PAIR>), dim3(2048), dim3(64), 0, 0, P, nb, cnt);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-28s %8.3f ms  (%.2f TB/s payload)\n", name, best, (double)nb * 65536 * 16 / (best * 1e-3) / 1e12);
    return 0;
}
// coalesced fill of one 1 MiB region per wave-iteration (k_predict_small's store shape), few waves per CU so that it fits beside k_scatter
__global__ void __launch_bounds__(64) k_fill_blocks(uint4 *p, uint32_t nblocks) {
    for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        uint4 *dst = p + (size_t)b * 65536u;
        for (uint32_t i = threadIdx.x; i < 65536u; i += 64u) dst[i] = make_uint4(i, b, 2, 3);
    }
}
int concurrent(uint4 *P, uint4 *Q, uint32_t nb, uint32_t *cnt, int fill_waves_per_cu) {
    hipStream_t sa, sb; CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, f1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&f1));
    float best = 1e9f, bestf = 0.f;
    for (int it = 0; it < 3; it++) {
        CHECK(hipMemset(cnt, 0, 4)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, sa)); CHECK(hipStreamWaitEvent(sb, e0, 0));
        hipLaunchKernelGGL(k_fill_blocks, dim3(256 * fill_waves_per_cu), dim3(64), 0, sb, Q, nb);
        hipLaunchKernelGGL((k_scatter<0, 0>), dim3(2048), dim3(64), 0, sa, P, nb, cnt);
        CHECK(hipEventRecord(f1, sb)); CHECK(hipStreamWaitEvent(sa, f1, 0));
        CHECK(hipEventRecord(e1, sa)); CHECK(hipEventSynchronize(e1));
        float ms, msf; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipEventElapsedTime(&msf, e0, f1));
        if (ms < best) { best = ms; bestf = msf; }
    }
    printf("scatter 1e9 x 16 B  ||  fill 16 GB with %2d waves per CU: both done after %7.3f ms (the fill after %7.3f ms)\n", fill_waves_per_cu, best, bestf);
    return 0;
}
int main() {
    const uint32_t nb = 15259;
    uint4 *P; uint32_t *cnt;
    CHECK(hipMalloc(&P, (size_t)nb * 65536 * 16));
    CHECK(hipMalloc(&cnt, 4));
    run<0, 0>(P, nb, cnt, "default");
    run<1, 0>(P, nb, cnt, "nt");
    run<2, 0>(P, nb, cnt, "sc0");
    run<3, 0>(P, nb, cnt, "sc1");
    run<4, 0>(P, nb, cnt, "sc0 sc1");
    run<5, 0>(P, nb, cnt, "sc0 sc1 nt");
    run<6, 0>(P, nb, cnt, "sc1 nt");
    run<7, 0>(P, nb, cnt, "sc0 nt");
    run<0, 1>(P, nb, cnt, "default, lane pairs adjacent");
    run<0, 2>(P, nb, cnt, "ascending ~6 apart");
    run<0, 3>(P, nb, cnt, "ascending ~6 apart, rotated");
    run<0, 4>(P, nb, cnt, "ascending ~37 apart");
    uint4 *Q; CHECK(hipMalloc(&Q, (size_t)nb * 65536 * 16));
    concurrent(P, Q, nb, cnt, 4);
    concurrent(P, Q, nb, cnt, 8);
    concurrent(P, Q, nb, cnt, 16);
    return 0;
}
