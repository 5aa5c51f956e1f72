// w3_sweep.h — the reference's parameter-sweep driver (bin/ordern/main.rs:9-80) as ONE device launch: lanes = configurations x
// blocks, every lane runs the reference's bit loop (predict -> update -> encode, main.rs:71-77) for its OrderN(bits, align)
// on its block with the counting sink ACStats (helpers.rs:60-90) — no bytes are produced, the result is the bit count.
// Round 1 ran the 115 configurations one after the other through w3_encode_blocks (57 s for 20 MB); a lane's time is the
// serial chain of its block whatever the lane count, so all of them together take the time of one.
#pragma once
#include "w3_generic.h"

namespace w3 {

struct SweepCfg {
    uint8_t  bits, align, use_hash, pad;
    uint32_t hash_mask, hist_mask;
    uint64_t base, stride;     // Counter table of lane (cfg, block b): tables + base + b * stride
};

struct SweepArgs {
    const uint8_t *in; uint64_t n;
    uint32_t block_size, nblocks, waves_per_cfg, ncfg, first_cfg;
    const SweepCfg *cfg;       // [all configurations]
    uint8_t *tables;
    uint32_t *out_bits;        // [all configurations][nblocks]
};

// ArithmeticCoder::encode (arithmetic_coder.rs:41-65) into ACStats: write_bit counts 1 + rev_bits, inc_parity counts later
struct StatsEncoder {
    uint32_t x1 = 0u, x2 = 0xFFFFFFFFu, rev = 0u, bits = 0u;
    __device__ __forceinline__ void encode(uint32_t bit, uint32_t prob) {
        const uint32_t p32 = prob ? (prob << 16) : 1u;
        const uint32_t xmid = x1 + __umulhi(x2 - x1, p32);
        if (bit) x2 = xmid; else x1 = xmid + 1u;
        const uint32_t d = x1 ^ x2;
        const uint32_t nn = d ? (uint32_t)__builtin_clz(d) : 32u;
        if (nn) {
            bits += nn + rev; rev = 0u;
            if (nn == 32u) { x1 = 0u; x2 = 0xFFFFFFFFu; }
            else { x1 <<= nn; x2 = (x2 << nn) | ((1u << nn) - 1u); }
        }
        const uint32_t m = (uint32_t)__builtin_clz(~((x1 & ~x2) << 1));
        x1 = (x1 << m) & 0x7FFFFFFFu;
        x2 = (x2 << m) | 0x80000000u | ((1u << m) - 1u);
        rev += m;
    }
};

__global__ void __launch_bounds__(64) k_sweep_ordern(SweepArgs a) {
    const uint32_t c = a.first_cfg + blockIdx.x / a.waves_per_cfg;          // one configuration per wave: uniform table kind
    const uint32_t b = (blockIdx.x % a.waves_per_cfg) * 64u + threadIdx.x;
    if (b >= a.nblocks) return;
    const SweepCfg cf = a.cfg[c];
    const uint64_t off = (uint64_t)b * a.block_size;
    const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
    LeafParam lp;
    lp.bits = cf.bits; lp.align = cf.align; lp.hist = 0; lp.max_bits = 0; lp.frozen = 0; lp.use_hash = cf.use_hash; lp.kind = 0;
    lp.tbl_off = 0; lp.hash_mask = cf.hash_mask; lp.hist_mask = cf.hist_mask; lp.lut = nullptr;
    uint8_t *tbl = a.tables + cf.base + (uint64_t)b * cf.stride;
    const uint32_t amask = (1u << cf.align) - 1u;
    StatsEncoder enc;
    uint32_t hist = 0u, t = 0u;
    for (uint32_t i = 0; i < len; i++) {
        const uint32_t byte = a.in[off + i];
        for (int s = 7; s >= 0; s--) {
            const uint32_t ctx = t ? (((hist & cf.hist_mask) << cf.align) | (t & amask)) : 0u;   // ordern.rs:35-43; ctx starts at 0
            uint32_t *slot = leaf_slot(lp, tbl, ctx);
            const uint32_t cv = *slot;
            const uint32_t bit = (byte >> s) & 1u;
            *slot = counter_update_packed(cv, bit);
            enc.encode(bit, counter_p_packed(cv));
            hist = (hist << 1) | bit;
            t++;
        }
    }
    a.out_bits[(uint64_t)c * a.nblocks + b] = enc.bits;
}

}  // namespace w3
