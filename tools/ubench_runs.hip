// ubench_runs.hip — what a wavefront's 16-byte loads cost the address path when its 64 lanes read 64 / R different streams in runs of
// R consecutive 16-byte pieces (R = 1: the coder's M-wave today, lane = block, every lane its own 128-byte line; R = 4 / 8: 64 or
// 128 contiguous bytes per stream, the rest of the way to a coalesced load), alone and BESIDE the rank kernels' store pattern
// (16-byte stores scattered over a block's 1 MiB region, tools/ubench_mem3.hip).  Question behind it (profiles/r4_experiments/):
// step k's coder runs beside step k+1's rank kernels, which are bound by the per-CU cost of their scattered stores — how much of
// their slow-down (12.5 + 13.0 ms alone, 17.1 + 16.4 beside the coder) is the coder's 1.5e9 lane-sized loads on the same path?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// nstreams streams of `len` 16-byte pieces each; a wavefront owns 64 consecutive streams (the coder: 64 blocks per workgroup, one
// M-wave) and advances ALL of them by 8 pieces per iteration with 8 load instructions — slot q = 64 k + lane of instruction k reads
// piece (q / 64R) R + q % R of stream (q / R) % 64: R = 1 is one piece of every stream per instruction (64 lines), R = 8 all 8 pieces
// of 8 streams (8 lines).  PACE: dependent ALU work per iteration so that the loop runs at the coder's pace (a byte position per
// ~600 cycles) instead of flat out; 0 = flat out.
template <int R>
__global__ void __launch_bounds__(64) k_stream_loads(const uint4 *src, uint32_t len, uint32_t nstreams, uint32_t pace, uint32_t *sink) {
    const uint32_t lane = threadIdx.x;
    uint32_t acc = 0;
    const uint32_t s0 = blockIdx.x * 64u;
    const uint4 *p[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t q = (uint32_t)k * 64u + lane;
        const uint32_t st = s0 + (q / R) % 64u, piece = (q / (64u * R)) * R + q % R;
        p[k] = src + (size_t)(st < nstreams ? st : nstreams - 1u) * len + piece;
    }
    uint4 nxt[8];
#pragma unroll
    for (int k = 0; k < 8; k++) nxt[k] = p[k][0];
    for (uint32_t i = 0; i < len; i += 8u) {
        uint4 cur[8];
#pragma unroll
        for (int k = 0; k < 8; k++) cur[k] = nxt[k];
        const uint32_t in = i + 8u < len ? i + 8u : i;
#pragma unroll
        for (int k = 0; k < 8; k++) nxt[k] = p[k][in];
#pragma unroll
        for (int k = 0; k < 8; k++) acc += cur[k].x ^ cur[k].y ^ cur[k].z ^ cur[k].w;
        for (uint32_t w = 0; w < pace; w++) acc = acc * 1664525u + 1013904223u;   // dependent chain: ~10 cycles per step
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the rank kernels' store pattern (ubench_mem3.hip k_scatter<0, 0>)
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
            const uint32_t pos = (k * 40503u) & 0xFFFFu;
            dst[pos] = make_uint4(k, b, sl, r);
        }
    }
}

template <int R>
int run(const uint4 *src, uint32_t len, uint32_t nstreams, uint32_t pace, uint4 *P, uint32_t nb, uint32_t *cnt, uint32_t *sink, bool with_scatter) {
    hipStream_t sa, sb; CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, a1, b1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&a1)); CHECK(hipEventCreate(&b1));
    float best_l = 1e9f, best_s = 1e9f;
    for (int it = 0; it < 3; it++) {
        CHECK(hipMemset(cnt, 0, 4)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, sa)); CHECK(hipStreamWaitEvent(sb, e0, 0));
        hipLaunchKernelGGL((k_stream_loads<R>), dim3(nstreams / 64u), dim3(64), 0, sa, src, len, nstreams, pace, sink);
        CHECK(hipEventRecord(a1, sa));
        if (with_scatter) hipLaunchKernelGGL(k_scatter, dim3(2048), dim3(64), 0, sb, P, nb, cnt);
        CHECK(hipEventRecord(b1, sb));
        CHECK(hipDeviceSynchronize());
        float ml, ms; CHECK(hipEventElapsedTime(&ml, e0, a1)); CHECK(hipEventElapsedTime(&ms, e0, b1));
        if (ml < best_l) best_l = ml;
        if (ms < best_s) best_s = ms;
    }
    printf("loads in runs of %2d lanes (%4u wavefronts, pace %3u)%s: loads done after %7.3f ms", R, nstreams / 64u, pace, with_scatter ? " || scatter 1e9 x 16 B" : "                      ", best_l);
    if (with_scatter) printf(", scatter after %7.3f ms", best_s);
    printf("\n");
    return 0;
}

int main() {
    const uint32_t nstreams = 15296, len = 65536;   // the coder's shape at enwik9 size: 239 x 64 streams of 65,536 16-byte pieces (16 GB)
    const uint32_t nb = 15259;
    uint4 *src, *P; uint32_t *cnt, *sink;
    CHECK(hipMalloc(&src, (size_t)nstreams * len * 16)); CHECK(hipMalloc(&P, (size_t)nb * 65536 * 16)); CHECK(hipMalloc(&cnt, 4)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(src, 1, (size_t)nstreams * len * 16)); CHECK(hipMemset(P, 0, (size_t)nb * 65536 * 16));
    {   // the scatter alone
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int it = 0; it < 3; it++) {
            CHECK(hipMemset(cnt, 0, 4)); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(k_scatter, dim3(2048), dim3(64), 0, 0, P, nb, cnt); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("scatter 1e9 x 16 B alone: %7.3f ms\n", best);
    }
    for (uint32_t pace : {0u, 400u}) {
        for (int ws = 0; ws < 2; ws++) {
            if (run<1>(src, len, nstreams, pace, P, nb, cnt, sink, ws)) return 1;
            if (run<2>(src, len, nstreams, pace, P, nb, cnt, sink, ws)) return 1;
            if (run<4>(src, len, nstreams, pace, P, nb, cnt, sink, ws)) return 1;
            if (run<8>(src, len, nstreams, pace, P, nb, cnt, sink, ws)) return 1;
        }
    }
    return 0;
}
