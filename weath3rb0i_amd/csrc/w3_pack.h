// w3_pack.h — pack the per-block byte streams into one contiguous buffer.
// Every block stream is byte aligned (ACWriter::flush pads, io.rs:91-100), so
// packing is byte granular: wave prefix-scan of the lengths, then a
// compaction copy.  Lengths > 0 always (flush writes at least one byte).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace w3 {

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Single workgroup (1024 threads = 16 waves): exclusive scan of lens[nb] into
// offs[nb]; total[0] = sum.  nb is small (N / block_size), one pass suffices.
__global__ void __launch_bounds__(1024) k_scan_lens(const uint32_t *lens, uint64_t *offs, uint64_t *total, uint32_t nb) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        uint32_t i = base + tid;
        uint64_t v = i < nb ? lens[i] : 0;
        uint64_t inc = wave_incl_scan_u64(v);
        if (lane == 63) wave_sum[wid] = inc;
        __syncthreads();
        uint64_t wprefix = 0;
        for (int w = 0; w < wid; w++) wprefix += wave_sum[w];
        uint64_t carry = carry_s;
        if (i < nb) offs[i] = carry + wprefix + inc - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + wprefix + inc;
        __syncthreads();
    }
    if (tid == 0) total[0] = carry_s;
}

// One workgroup per block: copy stripe -> out + offs[b].  Source stripes are
// 16-byte aligned; the destination is arbitrary, so copy bytewise at the
// ragged head/tail and 16 B per lane in between when alignment allows.
__global__ void __launch_bounds__(256) k_pack(const uint8_t *stripes, uint64_t stripe_stride, const uint32_t *lens,
                                              const uint64_t *offs, uint8_t *out, uint64_t out_cap, uint32_t nb) {
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint8_t *src = stripes + (uint64_t)b * stripe_stride;
        const uint32_t len = lens[b];
        const uint64_t o = offs[b];
        if (o + len > out_cap) continue;  // host reports W3_E_NOSPACE from the total
        uint8_t *dst = out + o;
        // bytewise, coalesced: consecutive lanes -> consecutive bytes
        for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) dst[i] = src[i];
    }
}

}  // namespace w3
