// w3_coder.h — CODE phase of the two-phase encoder: one wavefront lane per
// block runs the serial arithmetic-coder recurrence (arithmetic_coder.rs:41-65)
// over the probabilities the predict phase left in P.  The coder state
// (x1, x2, pending-parity count, bit accumulator) lives in the lane's VGPRs;
// nothing but P, the input byte and the output stripe touches memory.
#pragma once
#include "w3_device.h"

namespace w3 {

struct CoderArgs {
    const uint8_t *in;
    uint64_t n;
    uint32_t block_size, nblocks;
    const uint4 *P;        // [n] 8 x u16 per input byte
    uint8_t *stripes;      // block-major output stripes
    uint32_t stripe_cap;
    uint32_t *out_len;     // [nblocks]
    uint32_t *overflow;
};

__global__ void __launch_bounds__(64) k_coder(CoderArgs a) {
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= a.nblocks) return;
    const uint64_t off = (uint64_t)b * a.block_size;
    const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
    const uint4 *Pb = a.P + off;
    const uint8_t *blk = a.in + off;
    Encoder enc;
    enc.init(a.stripes + (uint64_t)b * a.stripe_cap, a.stripe_cap);
    for (uint32_t i = 0; i < len; i++) {
        const uint4 pv = Pb[i];
        const uint32_t byte = blk[i];
        enc.encode((byte >> 7) & 1u, pv.x & 0xFFFFu);
        enc.encode((byte >> 6) & 1u, pv.x >> 16);
        enc.encode((byte >> 5) & 1u, pv.y & 0xFFFFu);
        enc.encode((byte >> 4) & 1u, pv.y >> 16);
        enc.encode((byte >> 3) & 1u, pv.z & 0xFFFFu);
        enc.encode((byte >> 2) & 1u, pv.z >> 16);
        enc.encode((byte >> 1) & 1u, pv.w & 0xFFFFu);
        enc.encode(byte & 1u, pv.w >> 16);
    }
    const uint32_t produced = enc.flush();
    a.out_len[b] = produced;
    if (produced > a.stripe_cap) atomicOr(a.overflow, 1u);
}

}  // namespace w3
