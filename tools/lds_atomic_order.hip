// Does ds_add_rtn_u32 serialise the lanes of one wavefront that hit the SAME LDS address in ascending lane order?
// (The ISA manual does not say.)  Every lane adds inc[lane] to tbl[key[lane]] and keeps the returned old value; the host
// replays the adds in lane order and compares.  Also times the instruction under realistic key distributions.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_order lds_atomic_order.hip && ./lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(64) k_order(const uint32_t *key, const uint32_t *inc, uint32_t *old, uint32_t *fin, int rounds, int tbl_words) {
    __shared__ uint32_t tbl[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) tbl[i] = 0u;
    __builtin_amdgcn_wave_barrier();
    const size_t base = (size_t)blockIdx.x * rounds * 64;
    for (int r = 0; r < rounds; r++) {
        const uint32_t k = key[base + (size_t)r * 64 + lane], v = inc[base + (size_t)r * 64 + lane];
        old[base + (size_t)r * 64 + lane] = __hip_atomic_fetch_add(&tbl[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < tbl_words; i += 64) fin[(size_t)blockIdx.x * tbl_words + i] = tbl[i];
}

// throughput: 8 atomics per round (one per 256-entry sub-table), like a rank round would issue
__global__ void __launch_bounds__(64) k_rate(const uint32_t *key, uint32_t *sink, int rounds) {
    __shared__ uint32_t tbl[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) tbl[i] = 0u;
    __builtin_amdgcn_wave_barrier();
    uint32_t acc = 0;
    const uint32_t k0 = key[(size_t)(blockIdx.x % 64) * 64 + lane];
    for (int r = 0; r < rounds; r++) {
        const uint32_t k = (k0 + (uint32_t)r * 7u) & 255u;
#pragma unroll
        for (int j = 0; j < 8; j++) acc += __hip_atomic_fetch_add(&tbl[j * 256 + ((k >> j) | (k << (8 - j) & 255u))], 1u + (acc & 1u) * 65535u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    sink[blockIdx.x * 64 + lane] = acc;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main() {
    const int waves = 512, rounds = 256, tblw = 2048;
    const size_t n = (size_t)waves * rounds * 64;
    std::vector<uint32_t> key(n), inc(n), old(n), fin((size_t)waves * tblw);
    // per wave a different key distribution: 1, 2, 4 ... distinct keys; same bank different address; text-like skew
    for (int w = 0; w < waves; w++)
        for (int r = 0; r < rounds; r++)
            for (int l = 0; l < 64; l++) {
                const size_t i = ((size_t)w * rounds + r) * 64 + l;
                uint32_t k;
                switch (w % 8) {
                case 0: k = 5; break;                                  // all lanes one address
                case 1: k = rnd() & 1; break;
                case 2: k = rnd() & 7; break;
                case 3: k = (rnd() & 3) * 32 + 1; break;               // same bank, 4 addresses
                case 4: k = rnd() & 255; break;
                case 5: k = (rnd() % 100 < 60) ? 32 : rnd() & 2047; break;   // one hot key + noise
                case 6: k = (uint32_t)(l / 4) + (rnd() & 1) * 1024; break;
                default: k = rnd() & 2047; break;
                }
                key[i] = k;
                inc[i] = (rnd() & 1) ? 0x10000u : 1u;
            }
    uint32_t *dk, *di, *dold, *dfin;
    hipMalloc(&dk, n * 4); hipMalloc(&di, n * 4); hipMalloc(&dold, n * 4); hipMalloc(&dfin, fin.size() * 4);
    hipMemcpy(dk, key.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(di, inc.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_order, dim3(waves), dim3(64), 0, 0, dk, di, dold, dfin, rounds, tblw);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(old.data(), dold, n * 4, hipMemcpyDeviceToHost); hipMemcpy(fin.data(), dfin, fin.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, bad_fin = 0; long first_bad = -1;
    for (int w = 0; w < waves; w++) {
        std::vector<uint32_t> t(tblw, 0u);
        for (int r = 0; r < rounds; r++)
            for (int l = 0; l < 64; l++) {
                const size_t i = ((size_t)w * rounds + r) * 64 + l;
                if (old[i] != t[key[i]]) { bad++; if (first_bad < 0) first_bad = (long)i; }
                t[key[i]] += inc[i];
            }
        for (int i = 0; i < tblw; i++) bad_fin += fin[(size_t)w * tblw + i] != t[i];
    }
    printf("lane-order check: %zu adds, %zu returned values differ from the lane-order replay (first at %ld), %zu final words differ\n", n, bad, first_bad, bad_fin);
    if (first_bad >= 0) {
        const size_t r0 = (size_t)first_bad / 64 * 64;
        printf("wave kind %d, round keys/olds:", (int)(first_bad / 64 / rounds % 8));
        for (int l = 0; l < 64; l++) printf(" %u:%x", key[r0 + l], old[r0 + l]);
        printf("\n");
    }
    // rate
    uint32_t *dsink; hipMalloc(&dsink, 8192 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1024, 2048, 4096, 8192}) {
        const int rr = 2000;
        hipLaunchKernelGGL(k_rate, dim3(grid), dim3(64), 0, 0, dk + 4 * rounds * 64, dsink, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate, dim3(grid), dim3(64), 0, 0, dk + 4 * rounds * 64, dsink, rr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rate: %d waves x %d rounds x 8 ds_add_rtn (random 8-bit keys): %.3f ms -> %.2f G wave-rounds/s\n", grid, rr, ms, (double)grid * rr / ms / 1e6);
    }
    return bad || bad_fin ? 1 : 0;
}
