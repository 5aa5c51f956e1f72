// Does L2 / Infinity Cache merge the 8 partial (16 B) writes that every 128-B line of a block's P region
// receives over the block's lifetime?  Each wave permutes 65536 x 16 B inside its own 1 MiB region, `reps` times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(64) k_perm(uint4 *buf, int reps, uint32_t mult) {
    uint4 *base = buf + (size_t)blockIdx.x * 65536;
    for (int r = 0; r < reps; r++)
        for (uint32_t e = threadIdx.x; e < 65536; e += 64) {
            const uint32_t pos = (e * mult + r) & 0xFFFFu;
            base[pos] = make_uint4(e, r, pos, 1);
        }
}
int main() {
    uint4 *buf; hipMalloc(&buf, (size_t)4096 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint32_t mult : {1u, 40503u, 8u * 5u + 1u}) {
        for (int grid : {4096, 2048, 1024, 512, 256, 128, 64}) {
            int reps = 4096 / grid * 2; if (reps < 2) reps = 2;
            hipLaunchKernelGGL(k_perm, dim3(grid), dim3(64), 0, 0, buf, 1, mult);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_perm, dim3(grid), dim3(64), 0, 0, buf, reps, mult);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double stores = (double)grid * reps * 1024;
            printf("mult=%6u grid=%5d (%4d MiB live): %8.3f ms, %7.2f ns per wave-store chip-wide, %.1f GB/s useful\n", mult, grid, grid, ms,
                   ms * 1e6 / stores, stores * 1024 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
