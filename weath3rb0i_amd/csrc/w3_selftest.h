// w3_selftest.h — on-device exhaustive check of the division-free Counter::p
// against the literal u64 formula of models/counter.rs:13-18.
#pragma once
#include "w3_device.h"

namespace w3 {
__global__ void __launch_bounds__(256) k_selftest_counter_p(unsigned long long *mismatches, uint32_t c0_lo, uint32_t c0_hi) {
    // grid-stride over (c0 in [c0_lo, c0_hi), c1 in [0, 65536))
    const uint64_t total = (uint64_t)(c0_hi - c0_lo) << 16;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c0 = c0_lo + (uint32_t)(i >> 16), c1 = (uint32_t)(i & 0xFFFFu);
        const uint64_t p = ((uint64_t)1 << 17) * ((uint64_t)c1 + 1) / ((uint64_t)c0 + c1 + 2);
        const uint32_t want = (uint32_t)((p >> 1) + (p & 1));
        if (counter_p(c0, c1) != want) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}
}  // namespace w3
