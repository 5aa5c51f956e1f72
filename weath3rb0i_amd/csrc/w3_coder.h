// w3_coder.h — CODE phase of the two-phase encoder: one wavefront lane per
// block runs the serial arithmetic-coder recurrence (arithmetic_coder.rs:41-65)
// over the probabilities the predict phase left in P.  The coder state lives
// in the lane's VGPRs; only P, the input byte and the output stripe touch
// memory.  The serial chain (block_size*8 dependent steps per lane) bounds the
// whole encoder, so k_coder_fast is written for minimum instructions per step:
//
//  * both renormalisation loops (:51-62) collapse into ONE shift by
//    s = n + m, found with two clz (n equal leading bits, then m E3 bits).
//  * pending-parity bits (ACWriter::rev_bits, io.rs:56,66-68,84-87) are never
//    counted.  The accumulator always ends in a "slot": a 0 that stands for the
//    next real output bit b, followed by one 1 per pending bit.  write_bit's
//    rule "b, then rev_bits copies of !b" is then a single add of b into the
//    accumulator (0111..1 + 1 = 1000..0).  After the update, the top s+1 bits
//    of x1 are exactly  b | rest(n-1) | 0 | 1^m, i.e. the bits to append plus
//    the new slot, so emission is: acc += x1>>31; acc = acc<<s | bfe(x1,31-s,s).
//  * bits leave the accumulator 32 at a time once per input byte, only from
//    above the slot (so a later carry never has to reach memory).
//  * a lane whose pending run outgrows the 64-bit accumulator (probability
//    ~2^-39 per E3 episode; adversarial inputs can force it) gives up and its
//    block is re-coded by k_coder (counted pending bits, any length).
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
    uint32_t *flags;       // [0] stripe overflow, [1] number of blocks in redo
    uint32_t *redo;        // fast coder: blocks to re-code; safe coder: list to process (or null = all)
    uint32_t n_redo;
    uint32_t acc_limit;    // fast coder: max bits held before a step (46 = 64 - 18)
    uint32_t *out_bits;    // [nblocks] ACStats bit count of each block (helpers.rs:60-90: written bits before the flush), or null
};

// Robust coder: counted pending bits (Encoder in w3_device.h).  Codes the
// blocks listed in redo[0..n_redo) or, when redo == nullptr, every block.
__global__ void __launch_bounds__(64) k_coder(CoderArgs a) {
    const uint32_t idx = blockIdx.x * 64u + threadIdx.x;
    uint32_t b;
    if (a.redo) { if (idx >= a.n_redo) return; b = a.redo[idx]; }
    else { if (idx >= a.nblocks) return; b = idx; }
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
    if (a.out_bits) a.out_bits[b] = enc.stats_bits();
    const uint32_t produced = enc.flush();
    a.out_len[b] = produced;
    if (produced > a.stripe_cap) atomicOr(&a.flags[0], 1u);
}

// ---------------------------------------------------------------------------
// fast coder
// ---------------------------------------------------------------------------
struct FastEnc {
    uint32_t x1, x2;
    uint64_t acc;   // low nb bits valid; ends with slot(0) + one 1 per pending bit
    uint32_t nb;
    uint32_t pos;   // bytes written
};

// P holds Counter::p values only: 1..65535 (models/counter.rs:13-18), never 0.
// With range >= 2^30 before the step, the new range is >= 2^14-1, so x1 != x2
// afterwards and s = n + m <= 18.
// bitmask = 0xFFFFFFFF when the coded bit is 1, else 0 (prepared off the critical path).
__device__ __forceinline__ void fast_step(FastEnc &e, uint32_t bitmask, uint32_t p32) {
    const uint32_t range = e.x2 - e.x1;
    const uint32_t xmid = e.x1 + __umulhi(range, p32);                                        // lerp, :112-116
    e.x1 = (e.x1 & bitmask) | ((xmid + 1u) & ~bitmask);                                       // :45-48, branch- and VCC-free (v_bfi)
    e.x2 = (xmid & bitmask) | (e.x2 & ~bitmask);
    const uint32_t n = (uint32_t)__builtin_clz(e.x1 ^ e.x2);                                  // loop 1 trip count (x1 != x2)
    const uint32_t u = ~(e.x1 & ~e.x2) & (0x7FFFFFFFu >> n);                                  // != 0 because s <= 18
    const uint32_t c = (uint32_t)__builtin_clz(u);                                            // s + 1 = n + (loop 2 trip count) + 1
    const uint32_t s = c - 1u;
    e.acc += e.x1 >> 31;                                                                      // b into the slot (0 when n == 0)
    e.acc = (e.acc << s) | __builtin_amdgcn_ubfe(e.x1, 32u - c, s);                           // rest | new slot | pending ones
    e.nb += s;
    e.x1 = (e.x1 << c) >> 1;                                                                  // == (x1 << s) & 0x7FFFFFFF
    e.x2 = ~((~e.x2 << c) >> 1);                                                              // == (x2 << s) | 0x80000000 | ones(s)
}

__device__ __forceinline__ uint32_t trailing_ones64(uint64_t v) {
    const uint64_t z = ~v;
    return z ? (uint32_t)(__ffsll((long long)z) - 1) : 64u;
}

__global__ void __launch_bounds__(64) k_coder_fast(CoderArgs a) {
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= a.nblocks) return;
    const uint64_t off = (uint64_t)b * a.block_size;
    const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
    const uint4 *Pb = a.P + off;
    const uint8_t *blk = a.in + off;
    uint8_t *out = a.stripes + (uint64_t)b * a.stripe_cap;
    const uint32_t cap = a.stripe_cap, limit = a.acc_limit;
    FastEnc e;
    e.x1 = 0u; e.x2 = 0xFFFFFFFFu; e.acc = 0ull; e.nb = 1u; e.pos = 0u;   // one slot: the first output bit
    bool failed = false;

    // operands are fetched 4 bytes ahead with UNCONDITIONAL loads (index clamped): hipcc waits vmcnt(0)
    // right after a load it has to branch around, which would expose one HBM latency per input byte
    const uint32_t last = len - 1u;
    uint4 pq[4]; uint32_t bq[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t ic = min((uint32_t)k, last); pq[k] = Pb[ic]; bq[k] = blk[ic]; }
    for (uint32_t i = 0; i < len; i++) {
        const uint4 cur = pq[0];
        const uint32_t cb = bq[0];
#pragma unroll
        for (int k = 0; k < 3; k++) { pq[k] = pq[k + 1]; bq[k] = bq[k + 1]; }
        { const uint32_t ic = min(i + 4u, last); pq[3] = Pb[ic]; bq[3] = blk[ic]; }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (__builtin_expect(e.nb > limit, 0)) {
                // accumulator nearly full: drain finalised bytes (those above the slot) one at a time
                const uint32_t pend = trailing_ones64(e.acc) + 1u;
#pragma unroll 1
                while (e.nb >= pend + 8u) {
                    const uint8_t v = (uint8_t)(e.acc >> (e.nb - 8u));
                    if (e.pos < cap) out[e.pos] = v;
                    e.pos += 1u;
                    e.nb -= 8u;
                }
                if (e.nb > limit) {  // pending run longer than the accumulator: hand the block to k_coder
                    failed = true;
                    e.acc = 0ull; e.nb = 1u;
                }
            }
            const uint32_t w = j < 2 ? cur.x : j < 4 ? cur.y : j < 6 ? cur.z : cur.w;
            const uint32_t p32 = (j & 1) ? (w & 0xFFFF0000u) : (w << 16);
            fast_step(e, (uint32_t)__builtin_amdgcn_sbfe((int)cb, 7 - j, 1), p32);
        }
        // once per input byte: move 32 finalised bits out (never the slot or the pending ones)
        const uint32_t lo = (uint32_t)e.acc;
        const uint32_t pend = (~lo ? (uint32_t)__builtin_ctz(~lo) : 32u) + 1u;   // slot + pending ones (>= 33: nothing to move)
        if (e.nb >= pend + 32u) {
            const uint32_t wv = (uint32_t)(e.acc >> (e.nb - 32u));
            if (e.pos + 4u <= cap) {
                const uint32_t be = __builtin_bswap32(wv);
                __builtin_memcpy(out + e.pos, &be, 4);
            }
            e.pos += 4u;
            e.nb -= 32u;
        }
    }
    // ArithmeticCoder::flush -> ACWriter::flush(x2) (arithmetic_coder.rs:67-71, io.rs:91-100):
    // first bit x2>>31 (= 1) resolves the slot and the pending bits, then x2's next bits pad to a byte
    if (a.out_bits && !failed) a.out_bits[b] = 8u * e.pos + e.nb - (trailing_ones64(e.acc) + 1u);   // ACStats: all bits but the slot and the pending ones
    e.acc += 1ull;
    const uint32_t idx = e.nb & 7u;
    if (idx) {
        const uint32_t k = 8u - idx;
        e.acc = (e.acc << k) | ((e.x2 << 1) >> (32u - k));
        e.nb += k;
    }
#pragma unroll 1
    while (e.nb >= 8u) {
        const uint8_t v = (uint8_t)(e.acc >> (e.nb - 8u));
        if (e.pos < cap) out[e.pos] = v;
        e.pos += 1u;
        e.nb -= 8u;
    }
    if (failed) {
        const uint32_t k = atomicAdd(&a.flags[1], 1u);
        a.redo[k] = b;
    } else {
        a.out_len[b] = e.pos;
        if (e.pos > cap) atomicOr(&a.flags[0], 1u);
    }
}

// ---------------------------------------------------------------------------
// k_coder_x2 — the serial chain split over TWO wavefronts per 64 blocks.
//
// A lone wave issues a dependent VALU instruction only every ~6.6 cycles (4 if
// independent), and at enwik9 size three quarters of the SIMDs are idle while
// the coder runs.  So the step is cut along its dependency structure:
//   X-wave: the (x1, x2) recurrence only — lerp, range update, the two clz, the
//           combined shift.  It publishes one token per step, (x1 after the
//           update, c = s + 1), into an LDS ring and never waits for memory
//           stores or output bookkeeping.
//   O-wave: everything derived from the tokens — slot/carry accumulator, guard,
//           32-bit flushes to the stripe, final ACWriter::flush.  None of it
//           feeds back into the recurrence.
// The two waves of a workgroup sit on different SIMDs of one CU and hand
// tokens over through a 64 KiB LDS ring (16 input bytes deep, produced and
// consumed 8 bytes at a time).  LDS executes one wave's instructions in order,
// so "tokens, then the counter" needs no wait on the producer side.
// ---------------------------------------------------------------------------
#define W3_X2_RING 16u          // ring depth in input bytes (power of two, two halves of 8)
#define W3_X2_SPIN_LIMIT (1u << 24)
#ifndef W3_X3_LAZY
#define W3_X3_LAZY 8            // s_sleep units between polls of the mix / output waves of k_coder_x3
#endif

// Progress words of the wave pipelines: release on the producer's store (everything it wrote to the ring is visible first),
// acquire on the consumer's poll (its ring reads come after).  On gfx950 both cost one s_waitcnt lgkmcnt on LDS — the LDS
// executes a wave's operations in order anyway — but the ordering no longer rests on that plus compiler barriers
// (ADVICE r1).  The assembly loops of k_coder_x4 order their own LDS operations explicitly (in-order DS + s_waitcnt).
__device__ __forceinline__ uint32_t lds_load_u32(const volatile uint32_t *p) {
    return __hip_atomic_load(const_cast<const uint32_t *>(p), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store_u32(volatile uint32_t *p, uint32_t v) {
    __hip_atomic_store(const_cast<uint32_t *>(p), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ void __launch_bounds__(128) k_coder_x2(CoderArgs a) {
    __shared__ uint2 tok[W3_X2_RING * 8u * 64u];   // [ring byte][bit][lane]
    __shared__ uint32_t fin_x2[64];
    __shared__ uint32_t sync_w[4];                  // [0] bytes produced, [1] bytes consumed, [2] abort
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t b = blockIdx.x * 64u + lane;
    const bool act = b < a.nblocks;
    const uint64_t off = (uint64_t)(act ? b : 0u) * a.block_size;
    const uint32_t len = act ? (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size) : 0u;
    uint32_t maxlen = len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, d, 64));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    if (threadIdx.x < 4) sync_w[threadIdx.x] = 0u;
    __syncthreads();
    volatile uint32_t *prod = &sync_w[0], *cons = &sync_w[1], *abortf = &sync_w[2];

    if (wave == 0) {
        // ------------------------------ X-wave ------------------------------
        const uint4 *Pb = a.P + off;
        const uint8_t *blk = a.in + off;
        uint32_t x1 = 0u, x2 = 0xFFFFFFFFu;
        // operands one 4-byte group ahead, with UNCONDITIONAL loads (index clamped): see k_coder_fast
        const uint32_t last = (act && len) ? len - 1u : 0u;
        uint4 nx[4]; uint32_t nbytes[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t ic = min((uint32_t)k, last); nx[k] = Pb[ic]; nbytes[k] = blk[ic]; }
        bool dead = false;
        for (uint32_t i = 0; i < maxlen && !dead; i += 4) {
            if ((i & 7u) == 0u && i >= W3_X2_RING) {   // ring slots of bytes [i, i+8) must have been consumed
                uint32_t spins = 0;
                while (lds_load_u32(cons) + 8u < i) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > W3_X2_SPIN_LIMIT || lds_load_u32(abortf)) { lds_store_u32(abortf, 1u); dead = true; break; }
                }
                if (dead) break;
            }
            uint4 cur[4]; uint32_t cb[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { cur[k] = nx[k]; cb[k] = nbytes[k]; }
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint32_t ic = min(i + 4u + k, last); nx[k] = Pb[ic]; nbytes[k] = blk[ic]; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i + k < len) {
                    uint2 *slot = tok + ((size_t)((i + k) & (W3_X2_RING - 1u)) * 8u) * 64u + lane;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t w = j < 2 ? cur[k].x : j < 4 ? cur[k].y : j < 6 ? cur[k].z : cur[k].w;
                        const uint32_t p32 = (j & 1) ? (w & 0xFFFF0000u) : (w << 16);
                        const uint32_t bitmask = (uint32_t)__builtin_amdgcn_sbfe((int)cb[k], 7 - j, 1);
                        const uint32_t xmid = x1 + __umulhi(x2 - x1, p32);
                        x1 = (x1 & bitmask) | ((xmid + 1u) & ~bitmask);
                        x2 = (xmid & bitmask) | (x2 & ~bitmask);
                        const uint32_t n = (uint32_t)__builtin_clz(x1 ^ x2);
                        const uint32_t u = ~(x1 & ~x2) & (0x7FFFFFFFu >> n);
                        const uint32_t c = (uint32_t)__builtin_clz(u);
                        slot[j * 64] = make_uint2(x1, c);
                        x1 = (x1 << c) >> 1;
                        x2 = ~((~x2 << c) >> 1);
                    }
                    if (i + k + 1u == len) fin_x2[lane] = x2;
                }
            }
            __asm__ volatile("" ::: "memory");
            lds_store_u32(prod, min(i + 4u, maxlen));   // after the tokens: LDS runs one wave's ops in order
        }
        return;
    }

    // -------------------------------- O-wave --------------------------------
    uint8_t *out = a.stripes + (uint64_t)(act ? b : 0u) * a.stripe_cap;
    const uint32_t cap = act ? a.stripe_cap : 0u, limit = a.acc_limit;
    uint64_t acc = 0ull; uint32_t nb = 1u, pos = 0u;
    bool failed = false, dead = false;
    for (uint32_t i = 0; i < maxlen && !dead; i += 8) {
        const uint32_t need = min(i + 8u, maxlen);
        uint32_t spins = 0;
        while (lds_load_u32(prod) < need) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > W3_X2_SPIN_LIMIT || lds_load_u32(abortf)) { lds_store_u32(abortf, 1u); dead = true; break; }
        }
        if (dead) break;
        __asm__ volatile("" ::: "memory");
#pragma unroll 1
        for (uint32_t k = 0; k < 8u; k++) {
            if (i + k < len) {
                const uint2 *slot = tok + ((size_t)((i + k) & (W3_X2_RING - 1u)) * 8u) * 64u + lane;
                uint2 t[8];
#pragma unroll
                for (int j = 0; j < 8; j++) t[j] = slot[j * 64];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (__builtin_expect(nb > limit, 0)) {
                        const uint32_t pend = trailing_ones64(acc) + 1u;
#pragma unroll 1
                        while (nb >= pend + 8u) {
                            const uint8_t v = (uint8_t)(acc >> (nb - 8u));
                            if (pos < cap) out[pos] = v;
                            pos += 1u; nb -= 8u;
                        }
                        if (nb > limit) { failed = true; acc = 0ull; nb = 1u; }
                    }
                    const uint32_t x1v = t[j].x, c = t[j].y, s = c - 1u;
                    acc += x1v >> 31;
                    acc = (acc << s) | __builtin_amdgcn_ubfe(x1v, 32u - c, s);
                    nb += s;
                }
                const uint32_t lo = (uint32_t)acc;
                const uint32_t pend = (~lo ? (uint32_t)__builtin_ctz(~lo) : 32u) + 1u;
                if (nb >= pend + 32u) {
                    const uint32_t wv = (uint32_t)(acc >> (nb - 32u));
                    if (pos + 4u <= cap) { const uint32_t be = __builtin_bswap32(wv); __builtin_memcpy(out + pos, &be, 4); }
                    pos += 4u; nb -= 32u;
                }
            }
        }
        __asm__ volatile("" ::: "memory");
        lds_store_u32(cons, need);
    }
    if (dead) { if (lane == 0) atomicOr(&a.flags[0], 2u); return; }
    if (!act) return;
    // ACWriter::flush(x2) (io.rs:91-100)
    const uint32_t x2f = fin_x2[lane];
    if (a.out_bits && !failed) a.out_bits[b] = 8u * pos + nb - (trailing_ones64(acc) + 1u);   // ACStats: all bits but the slot and the pending ones
    acc += 1ull;
    const uint32_t idx = nb & 7u;
    if (idx) { const uint32_t k = 8u - idx; acc = (acc << k) | ((x2f << 1) >> (32u - k)); nb += k; }
#pragma unroll 1
    while (nb >= 8u) {
        const uint8_t v = (uint8_t)(acc >> (nb - 8u));
        if (pos < cap) out[pos] = v;
        pos += 1u; nb -= 8u;
    }
    if (failed) { const uint32_t k = atomicAdd(&a.flags[1], 1u); a.redo[k] = b; }
    else { a.out_len[b] = pos; if (pos > cap) atomicOr(&a.flags[0], 1u); }
}

}  // namespace w3

namespace w3 {

// ---------------------------------------------------------------------------
// k_coder_x3<L> — three wavefronts per 64 blocks: MIX -> RECURRENCE -> OUTPUT.
//
//   M-wave: streams the L leaves' probability streams and the input bytes from
//           HBM (the only wave that touches global loads), applies OpinionMixer2
//           (leftmost leaf of maximal |p-1/2|) and hands the X-wave, per step,
//           the two operands it needs: p32 = p<<16 and the bit as a 0/~0 mask.
//           This replaces the separate k_mix pass (48+16 GB of traffic per GB
//           of input at L = 3) and keeps memory latency out of the recurrence.
//   X-wave: (x1, x2) recurrence only, LDS in, LDS out: 17 instructions, 9 deep (comment at the wave).
//   O-wave: accumulator, flushes, final ACWriter::flush (see k_coder_x2).
// Two LDS rings of 16 input bytes each (M->X operands, X->O tokens): 128 KiB.
// ---------------------------------------------------------------------------
struct Coder3Args {
    const uint8_t *in;
    uint64_t n;
    uint32_t block_size, nblocks;
    const uint4 *src[4];   // leaf streams in leaf order (L of them)
    uint8_t *stripes;
    uint32_t stripe_cap;
    uint32_t *out_len;
    uint32_t *flags;       // [0] bit0 stripe overflow, bit1 pipeline timeout; [1] blocks in redo
    uint32_t *redo;
    uint32_t acc_limit;
    uint32_t *out_bits;    // [nblocks] ACStats bit count of each block (helpers.rs:60-90), or null
    uint32_t prio_mo;      // k_coder_x5: the mix and output waves run at s_setprio 2 (experiment, W3_OPT_TUNE bit 7)
};

// SLEEP: s_sleep units (64 clocks) between polls.  The recurrence wave polls eagerly (it is the critical path); the mix and
// output waves wait for it half a ring (8 bytes = 64 steps, ~9,000 clocks) at a time, and their polls are LDS reads that
// queue up in front of the recurrence wave's ring accesses: at s_sleep 1 they were 3x its own LDS traffic.
template <int SLEEP = 1>
__device__ __forceinline__ uint32_t spin_until_ge(const volatile uint32_t *ctr, uint32_t need, volatile uint32_t *abortf, bool &dead) {
    uint32_t v = lds_load_u32(ctr), spins = 0;
    while (v < need) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if (++spins > W3_X2_SPIN_LIMIT || lds_load_u32(abortf)) { lds_store_u32(abortf, 1u); dead = true; break; }
        v = lds_load_u32(ctr);
    }
    return v;
}

template <int L>
__global__ void __launch_bounds__(192) k_coder_x3(Coder3Args a) {
    __shared__ uint2 opq[W3_X2_RING * 8u * 64u];   // M -> X: (p32, bitmask) per step   [ring byte][bit][lane]
    __shared__ uint2 tok[W3_X2_RING * 8u * 64u];   // X -> O: (x1 after update, s)      [ring byte][bit][lane]
    __shared__ uint32_t fin_x2[64];
    __shared__ uint32_t sync_w[8];                 // [0] M produced, [1] X consumed, [2] X produced, [3] O consumed, [4] abort
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t b = blockIdx.x * 64u + lane;
    const bool act = b < a.nblocks;
    const uint64_t off = (uint64_t)(act ? b : 0u) * a.block_size;
    const uint32_t len = act ? (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size) : 0u;
    uint32_t maxlen = len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, d, 64));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    if (threadIdx.x < 8) sync_w[threadIdx.x] = 0u;
    __syncthreads();
    volatile uint32_t *m_prod = &sync_w[0], *x_cons = &sync_w[1], *x_prod = &sync_w[2], *o_cons = &sync_w[3], *abortf = &sync_w[4];
    bool dead = false;

    if (wave == 0) {
        // ------------------------------ M-wave ------------------------------
        const uint32_t last = (act && len) ? len - 1u : 0u;
        const uint8_t *blk = a.in + off;
        uint4 nx[L][4]; uint32_t nbyte[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t ic = min((uint32_t)k, last);
            nbyte[k] = blk[ic];
#pragma unroll
            for (int l = 0; l < L; l++) nx[l][k] = a.src[l][off + ic];
        }
        uint32_t seen = 0;   // last value read from x_cons
        for (uint32_t i = 0; i < maxlen && !dead; i += 4) {
            if ((i & 7u) == 0u && i + 8u > seen + W3_X2_RING) {   // ring slots of bytes [i, i+8) must have been consumed
                seen = spin_until_ge<W3_X3_LAZY>(x_cons, i + 8u - W3_X2_RING, abortf, dead);
                if (dead) break;
            }
            uint4 cur[L][4]; uint32_t cb[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                cb[k] = nbyte[k];
#pragma unroll
                for (int l = 0; l < L; l++) cur[l][k] = nx[l][k];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t ic = min(i + 4u + k, last);
                nbyte[k] = blk[ic];
#pragma unroll
                for (int l = 0; l < L; l++) nx[l][k] = a.src[l][off + ic];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i + k < len) {
                    uint2 *slot = opq + ((size_t)((i + k) & (W3_X2_RING - 1u)) * 8u) * 64u + lane;
#pragma unroll
                    for (int q = 0; q < 4; q++) {   // one dword = two steps
                        uint32_t w0 = q == 0 ? cur[0][k].x : q == 1 ? cur[0][k].y : q == 2 ? cur[0][k].z : cur[0][k].w;
                        if constexpr (L > 1) {
                            // OpinionMixer2 for both steps of the dword at once with packed 16-bit VALU (the M-wave, not the
                            // recurrence, was the slowest wave for L >= 2: a lone wave pays per instruction).  Distances
                            // |p - 32768| <= 32767 because Counter::p is in [1, 65535], so their difference fits an i16 and its
                            // sign is the "strictly farther" mask (ties keep the left leaf).
                            u16x2 P = as_u16x2(w0), D = pk_opinion_dist(P);
#pragma unroll
                            for (int l = 1; l < L; l++) {
                                const uint32_t w = q == 0 ? cur[l][k].x : q == 1 ? cur[l][k].y : q == 2 ? cur[l][k].z : cur[l][k].w;
                                const u16x2 Q = as_u16x2(w), E = pk_opinion_dist(Q);
                                const uint32_t mask = pk_farther_mask(D, E);                   // 0xFFFF where E > D
                                P = as_u16x2((as_u32(Q) & mask) | (as_u32(P) & ~mask));
                                D = __builtin_elementwise_max(D, E);
                            }
                            w0 = as_u32(P);
                        }
                        const uint32_t plo = w0 & 0xFFFFu, phi = w0 >> 16;
                        slot[(2 * q) * 64] = make_uint2(plo << 16, (uint32_t)__builtin_amdgcn_sbfe((int)cb[k], 7 - 2 * q, 1));
                        slot[(2 * q + 1) * 64] = make_uint2(phi << 16, (uint32_t)__builtin_amdgcn_sbfe((int)cb[k], 6 - 2 * q, 1));
                    }
                }
            }
            __asm__ volatile("" ::: "memory");
            lds_store_u32(m_prod, min(i + 4u, maxlen));
        }
        return;
    }

    if (wave == 1) {
        // ------------------------------ X-wave ------------------------------
        // The recurrence (arithmetic_coder.rs:41-65) as a 9-deep dependency chain of 15 VALU instructions per step.
        // (Measured: the shorter chain bought nothing — 30.2 -> 29.8 ms — because ONE wave issues a VALU instruction only
        // every ~7-8 cycles whether it is dependent or not; the step costs its instruction COUNT, ~20 with the LDS ops.)
        //   m    = mulhi(d, p32)                                  d = x2 - x1, carried from the previous step
        //   x2n  = bit ? x1 + m : x2 ;  x1n = bit ? x1 : x1 + m + 1            (1 takes the low sub-interval, :45-48)
        //   s    = clz((x1n ^ x2n) & ((~x1n | x2n) << 1 | 1))     both renormalisation loops in ONE count: n equal
        //          leading bits, then the m E3 positions where x1n = 1 and x2n = 0 (position q survives the mask iff
        //          x1n, x2n differ at q and q+1 is not an E3 position); s = n + m <= 18 because p is a Counter::p
        //   X1   = x1n << s ;  X2 = ((x2n + 1) << s) - 1          the raw shifts; the top-bit fix-ups of loop 2
        //          (x1 &= 0x7FFFFFFF, x2 |= 0x80000000, :59-60) swap a (1, 0) pair of top bits or do nothing, so
        //          d = X2 - X1 (mod 2^32) needs neither and they run beside the next step's mulhi
        uint32_t x1 = 0u, x2 = 0xFFFFFFFFu, d = 0xFFFFFFFFu;
        uint32_t seen_m = 0, seen_o = 0;
        for (uint32_t i = 0; i < maxlen && !dead; i += 8) {
            const uint32_t need = min(i + 8u, maxlen);
            if (seen_m < need) { seen_m = spin_until_ge(m_prod, need, abortf, dead); if (dead) break; }
            if (i + 8u > seen_o + W3_X2_RING) { seen_o = spin_until_ge(o_cons, i + 8u - W3_X2_RING, abortf, dead); if (dead) break; }
            __asm__ volatile("" ::: "memory");
#pragma unroll 1
            for (uint32_t k = 0; k < 8u; k++) {
                if (i + k < len) {
                    const size_t ring = ((size_t)((i + k) & (W3_X2_RING - 1u)) * 8u) * 64u + lane;
                    // token slot base as an LDS byte address in a register: the ring sits above 64 KiB, beyond the reach of the
                    // 16-bit ds offset field, and hipcc re-forms base + constant with an extra VALU instruction per store otherwise
                    typedef __attribute__((address_space(3))) uint64_t lds_tok_t;
                    lds_tok_t *tokp = (lds_tok_t *)&tok[ring];
                    asm volatile("" : "+v"(tokp));
                    uint2 op[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) op[j] = opq[ring + j * 64];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t p32 = op[j].x, bitmask = op[j].y;   // ~0 when the coded bit is 1
                        // The whole step as ONE asm statement (15 VALU): hipcc pads every separate inline-asm VALU statement with an
                        // s_nop (an issue slot of the lone wave each), and its own rendering of the C form takes 22 instructions.
                        //   m = mulhi(d, p32); xmid = x1 + m; xmid1 = xmid + 1
                        //   x2n = bit ? xmid : x2;  x1n = bit ? x1 : xmid1                      (v_bfi: (mask & a) | (~mask & b))
                        //   u = (x1n ^ x2n) & ((~x1n | x2n) << 1 | 1);  s = clz(u)
                        //   X1 = x1n << s;  X2 = ((x2n + 1) << s) - 1;  d = X2 - X1;  x1 = X1 & 0x7FFFFFFF;  x2 = X2 | 0x80000000
                        uint32_t x1n, sft, t0, t1;
                        asm("v_mul_hi_u32 %4, %2, %7\n\t"          // t0 = m
                            "v_add_u32 %5, %0, %4\n\t"             // t1 = xmid
                            "v_add3_u32 %4, %0, %4, 1\n\t"         // t0 = xmid1
                            "v_bfi_b32 %1, %8, %5, %1\n\t"         // x2 = x2n
                            "v_bfi_b32 %3, %8, %0, %4\n\t"         // x1n
                            "v_bfi_b32 %4, %3, %1, -1\n\t"         // t0 = ~x1n | x2n
                            "v_lshl_or_b32 %4, %4, 1, 1\n\t"       // t0 = g
                            "v_bitop3_b32 %5, %4, %3, %1 bitop3:0x60\n\t"   // t1 = u = g & (x1n ^ x2n)
                            "v_ffbh_u32 %6, %5\n\t"                // s
                            "v_add_u32 %4, 1, %1\n\t"              // t0 = x2n + 1
                            "v_lshlrev_b32 %0, %6, %3\n\t"         // x1 = X1 (raw)
                            "v_lshl_add_u32 %1, %4, %6, -1\n\t"    // x2 = X2 (raw)
                            "v_sub_u32 %2, %1, %0\n\t"             // d
                            "v_and_b32 %0, 0x7fffffff, %0\n\t"
                            "v_or_b32 %1, 0x80000000, %1"
                            : "+v"(x1), "+v"(x2), "+v"(d), "=&v"(x1n), "=&v"(t0), "=&v"(t1), "=&v"(sft)
                            : "v"(p32), "v"(bitmask));
                        tokp[j * 64] = ((uint64_t)sft << 32) | x1n;
                    }
                    if (i + k + 1u == len) fin_x2[lane] = x2;
                }
            }
            __asm__ volatile("" ::: "memory");
            lds_store_u32(x_cons, need);
            lds_store_u32(x_prod, need);
        }
        return;
    }

    // -------------------------------- O-wave --------------------------------
    uint8_t *out = a.stripes + (uint64_t)(act ? b : 0u) * a.stripe_cap;
    const uint32_t cap = act ? a.stripe_cap : 0u, limit = a.acc_limit;
    uint64_t acc = 0ull; uint32_t nb = 1u, pos = 0u;
    bool failed = false;
    uint32_t seen_x = 0;
    for (uint32_t i = 0; i < maxlen && !dead; i += 8) {
        const uint32_t need = min(i + 8u, maxlen);
        if (seen_x < need) { seen_x = spin_until_ge<W3_X3_LAZY>(x_prod, need, abortf, dead); if (dead) break; }
        __asm__ volatile("" ::: "memory");
#pragma unroll 1
        for (uint32_t k = 0; k < 8u; k++) {
            if (i + k < len) {
                const uint2 *slot = tok + ((size_t)((i + k) & (W3_X2_RING - 1u)) * 8u) * 64u + lane;
                uint2 t[8];
#pragma unroll
                for (int j = 0; j < 8; j++) t[j] = slot[j * 64];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (__builtin_expect(nb > limit, 0)) {
                        const uint32_t pend = trailing_ones64(acc) + 1u;
#pragma unroll 1
                        while (nb >= pend + 8u) {
                            const uint8_t v = (uint8_t)(acc >> (nb - 8u));
                            if (pos < cap) out[pos] = v;
                            pos += 1u; nb -= 8u;
                        }
                        if (nb > limit) { failed = true; acc = 0ull; nb = 1u; }
                    }
                    const uint32_t x1v = t[j].x, s = t[j].y;   // token = (x1 after the update, shift count)
                    acc += x1v >> 31;
                    acc = (acc << s) | __builtin_amdgcn_ubfe(x1v, 31u - s, s);
                    nb += s;
                }
                const uint32_t lo = (uint32_t)acc;
                const uint32_t pend = (~lo ? (uint32_t)__builtin_ctz(~lo) : 32u) + 1u;
                if (nb >= pend + 32u) {
                    const uint32_t wv = (uint32_t)(acc >> (nb - 32u));
                    if (pos + 4u <= cap) { const uint32_t be = __builtin_bswap32(wv); __builtin_memcpy(out + pos, &be, 4); }
                    pos += 4u; nb -= 32u;
                }
            }
        }
        __asm__ volatile("" ::: "memory");
        lds_store_u32(o_cons, need);
    }
    if (dead) { if (lane == 0) atomicOr(&a.flags[0], 2u); return; }
    if (!act) return;
    const uint32_t x2f = fin_x2[lane];
    if (a.out_bits && !failed) a.out_bits[b] = 8u * pos + nb - (trailing_ones64(acc) + 1u);   // ACStats: all bits but the slot and the pending ones
    acc += 1ull;
    const uint32_t idx = nb & 7u;
    if (idx) { const uint32_t k = 8u - idx; acc = (acc << k) | ((x2f << 1) >> (32u - k)); nb += k; }
#pragma unroll 1
    while (nb >= 8u) {
        const uint8_t v = (uint8_t)(acc >> (nb - 8u));
        if (pos < cap) out[pos] = v;
        pos += 1u; nb -= 8u;
    }
    if (failed) { const uint32_t k = atomicAdd(&a.flags[1], 1u); a.redo[k] = b; }
    else { a.out_len[b] = pos; if (pos > cap) atomicOr(&a.flags[0], 1u); }
}

}  // namespace w3
