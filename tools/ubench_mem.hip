// ubench_mem.hip — cost of divergent global stores per wave instruction (what bounds the partitioned predict kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// each wave owns a 1 MiB region; per iteration every lane stores W bytes at a position chosen by MODE:
// 0 = coalesced (lane-contiguous), 1 = 16 runs of 4 lanes, 2 = fully scattered (64 different 128-B lines)
template <int W, int MODE>
__global__ void __launch_bounds__(64) k_st(uint8_t *buf, int iters) {
    const uint32_t lane = threadIdx.x;
    uint8_t *base = buf + (size_t)blockIdx.x * (1u << 20);
    uint32_t x = blockIdx.x * 977u + 1u;
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        uint32_t off;
        if (MODE == 0) off = ((x >> 8) & 0x3FFu) * 1024u % (1u << 20) + lane * W;
        else if (MODE == 1) off = (((x >> 6) + (lane >> 2) * 4099u) & 0x1FFFu) * 128u % (1u << 20) + (lane & 3u) * W;
        else off = (((x >> 6) + lane * 4099u) & 0x1FFFu) * 128u % (1u << 20);
        off &= ~(uint32_t)(W - 1);
        if (W == 4) *reinterpret_cast<uint32_t *>(base + off) = x;
        if (W == 8) *reinterpret_cast<uint2 *>(base + off) = make_uint2(x, i);
        if (W == 16) *reinterpret_cast<uint4 *>(base + off) = make_uint4(x, i, lane, 7);
    }
}
template <int W, int MODE>
int run(uint8_t *buf, int grid, const char *name) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    hipLaunchKernelGGL((k_st<W, MODE>), dim3(grid), dim3(64), 0, 0, buf, 100);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_st<W, MODE>), dim3(grid), dim3(64), 0, 0, buf, iters);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double per_cu_ns = ms * 1e6 / ((double)iters * grid / 256.0);
    printf("%-34s W=%2d grid=%5d: %8.3f ms  %7.1f ns per wave-store per CU  (%.2f GB/s useful)\n", name, W, grid, ms, per_cu_ns,
           (double)iters * grid * 64 * W / (ms * 1e-3) / 1e9);
    return 0;
}
int main() {
    uint8_t *buf; const int grid = 4096;
    CHECK(hipMalloc(&buf, (size_t)grid << 20));
    run<8, 0>(buf, grid, "coalesced 8B");
    run<8, 1>(buf, grid, "16 runs x 4 lanes, 8B");
    run<8, 2>(buf, grid, "64 lines scattered, 8B");
    run<16, 2>(buf, grid, "64 lines scattered, 16B");
    run<4, 2>(buf, grid, "64 lines scattered, 4B");
    run<16, 2>(buf, 1024, "64 lines scattered, 16B");
    run<16, 0>(buf, grid, "coalesced 16B");
    return 0;
}
