// ubench_write_bw.hip — streaming WRITE bandwidth of the chip (what bounds the predict kernels): 16 GB written once, coalesced,
// in the shapes the kernels use; plus a read and a copy for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_fill16(uint4 *p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
// one wavefront per 1 MiB block region, 1 KiB per wave-store (k_predict_small's shape)
__global__ void __launch_bounds__(64) k_fill_blocks(uint4 *p, uint32_t nblocks) {
    for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        uint4 *dst = p + (size_t)b * 65536u;
        for (uint32_t i = threadIdx.x; i < 65536u; i += 64u) dst[i] = make_uint4(i, b, 2, 3);
    }
}
__global__ void __launch_bounds__(256) k_read16(const uint4 *p, size_t n16, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_copy16(const uint4 *s, uint4 *d, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
int main() {
    const uint32_t nb = 15259; const size_t n16 = (size_t)nb * 65536;   // 16 GB
    uint4 *a, *b; uint32_t *o;
    CHECK(hipMalloc(&a, n16 * 16)); CHECK(hipMalloc(&b, n16 * 16)); CHECK(hipMalloc(&o, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char *name, int kind, double gb) -> int {
        float best = 1e9f;
        for (int it = 0; it < 3; it++) {
            CHECK(hipEventRecord(e0, 0));
            if (kind == 0) hipLaunchKernelGGL(k_fill16, dim3(256 * 16), dim3(256), 0, 0, a, n16);
            if (kind == 1) hipLaunchKernelGGL(k_fill_blocks, dim3(256 * 20), dim3(64), 0, 0, a, nb);
            if (kind == 2) hipLaunchKernelGGL(k_read16, dim3(256 * 16), dim3(256), 0, 0, a, n16, o);
            if (kind == 3) hipLaunchKernelGGL(k_copy16, dim3(256 * 16), dim3(256), 0, 0, a, b, n16);
            if (kind == 4) CHECK(hipMemsetAsync(a, 0x5a, n16 * 16, 0));
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-44s %8.3f ms  %6.2f TB/s\n", name, best, gb / best);
        return 0;
    };
    run("fill 16 GB, 16 B per thread, grid-stride", 0, 16.0);
    run("fill 16 GB, wave per 1 MiB region", 1, 16.0);
    run("hipMemsetAsync 16 GB", 4, 16.0);
    run("read 16 GB", 2, 16.0);
    run("copy 16 GB (16 read + 16 written)", 3, 32.0);
    return 0;
}
