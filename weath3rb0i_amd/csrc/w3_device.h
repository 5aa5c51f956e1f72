// w3_device.h — device-side primitives of the weath3rb0i hot path for gfx950.
// Integer-only, bit-exact restatements for one wavefront lane; every function
// cites the reference file:line (under /root/reference/src) whose arithmetic
// it reproduces.  No CUDA paths, no portability layer: CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace w3 {

// ---------------------------------------------------------------------------
// Counter  (models/counter.rs:4-26)
// ---------------------------------------------------------------------------
// p = round_half_up( 2^17*(c1+1) / (c0+c1+2) ) without a 64-bit divide:
// p = ((q >> 1) + (q & 1)) with q = floor(2^17 a / d) (counter.rs:10-18)  ==  floor(2^16 a / d + 1/2)  ==  floor((2^17 a + d) / 2d),
// a = c1 + 1, d = c0 + c1 + 2.  f32 estimate biased low (x + 0.47 with |error| < 0.02: never above the true value, at most
// one below), then ONE exact integer correction: R = 2^17 a + d - p * 2d lies in [0, 4d); p += R >= 2d.  R is tiny, so
// mod-2^32 arithmetic is exact, and p < 2^17, 2d < 2^19 keep the product in the full-rate 24-bit multiplier.
// Exact for every reachable Counter state (c0, c1 <= 65535): verified exhaustively on the device (w3_selftest_counter_p,
// tests/test_gpu_parity.py::test_counter_p_exhaustive).
__device__ __forceinline__ uint32_t counter_p(uint32_t c0, uint32_t c1) {
    const uint32_t d = c0 + c1 + 2u;
    const uint32_t a = c1 + 1u;
    const float t = (float)a * __builtin_amdgcn_rcpf((float)d);
    uint32_t p = (uint32_t)__builtin_fmaf(t, 65536.0f, 0.47f);
    const uint32_t d2 = d + d;
    const uint32_t R = ((a << 17) + d) - __umul24(p, d2);
    return p + (R >= d2 ? 1u : 0u);
}

// packed counter: low 16 = data[0], high 16 = data[1]
__device__ __forceinline__ uint32_t counter_p_packed(uint32_t c) { return counter_p(c & 0xFFFFu, c >> 16); }

__device__ __forceinline__ uint32_t counter_update_packed(uint32_t c, uint32_t bit) {  // counter.rs:20-26
    uint32_t c0 = c & 0xFFFFu, c1 = c >> 16;
    if (bit) c1 += 1u; else c0 += 1u;
    if ((bit ? c1 : c0) == 0xFFFFu) {
        c0 = (c0 >> 1) + (c0 & 1u);
        c1 = (c1 >> 1) + (c1 & 1u);
    }
    return c0 | (c1 << 16);
}

// OpinionMixer2::mix distance (mixers/opinion_mixer2.rs:5-10).  A BestOfTwo
// tree of any shape returns the LEFTMOST leaf (in-order) of maximal distance,
// because mix() keeps p1 on ties; callers scan leaves left to right with '>'.
__device__ __forceinline__ uint32_t opinion_dist(uint32_t p) { return p >= 32768u ? p - 32768u : 32768u - p; }
// the same for two u16 probabilities per dword (packed 16-bit VALU)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ i16x2 as_i16x2(u16x2 v) { return __builtin_bit_cast(i16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t as_u32(i16x2 v) { return __builtin_bit_cast(uint32_t, v); }
// |p - 32768| of two u16 probabilities at once (mixers/opinion_mixer2.rs:5-10)
__device__ __forceinline__ u16x2 pk_opinion_dist(u16x2 p) {
    const u16x2 half = {32768, 32768};
    return __builtin_elementwise_max(p, half) - __builtin_elementwise_min(p, half);
}
// 0xFFFF in the halves where e > d (d, e: packed distances <= 32767).  As asm: hipcc turns `as_i16x2(d - e) >> 15` into a compare
// and a select per half (26 instead of 17 instructions per leaf pair and dword in the mixing loops).
__device__ __forceinline__ uint32_t pk_farther_mask(u16x2 d, u16x2 e) {
    uint32_t m;
    asm("v_pk_sub_i16 %0, %1, %2\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=&v"(m) : "v"(as_u32(d)), "v"(as_u32(e)));
    return m;
}


// ---------------------------------------------------------------------------
// Bit sink of one lane: ACWriter (entropy_coding/io.rs:52-101) with the
// per-bit packing replaced by a 64-bit accumulator flushed 32 bits at a time.
// ---------------------------------------------------------------------------
struct BitSink {
    uint64_t acc;   // newest bit at LSB; only the low `nb` bits are meaningful
    uint32_t nb;    // < 32 between calls
    uint8_t *base;  // 4-byte aligned stripe of this lane
    uint32_t pos;   // bytes produced so far (keeps counting past cap)
    uint32_t cap;

    __device__ __forceinline__ void init(uint8_t *b, uint32_t c) { acc = 0; nb = 0; base = b; pos = 0; cap = c; }

    __device__ __forceinline__ void put(uint32_t val, uint32_t k) {  // k in 0..32, val < 2^k
        acc = (acc << k) | val;
        nb += k;
        if (nb >= 32u) {
            uint32_t w = (uint32_t)(acc >> (nb - 32u));
            if (pos + 4u <= cap) *reinterpret_cast<uint32_t *>(base + pos) = __builtin_bswap32(w);
            pos += 4u;
            nb -= 32u;
        }
    }
    // after the stream is byte aligned: write the nb/8 tail bytes
    __device__ __forceinline__ void finish() {
        while (nb >= 8u) {
            uint8_t b = (uint8_t)(acc >> (nb - 8u));
            if (pos < cap) base[pos] = b;
            pos += 1u;
            nb -= 8u;
        }
    }
};

// ---------------------------------------------------------------------------
// ArithmeticCoder, encoder side (entropy_coding/arithmetic_coder.rs:37-71,109-119)
// The two renormalisation while-loops are collapsed with clz:
//   loop 1 (:51-55) runs n = clz(x1^x2) times and emits the top n bits of x1;
//   loop 2 (:58-62, E3) runs m = clo(((x1 & ~x2) << 1)) times.
// write_bit's pending-parity rule (io.rs:70-89): first bit b, then rev_bits
// copies of !b, then the rest.
// ---------------------------------------------------------------------------
struct Encoder {
    uint32_t x1, x2, rev;
    BitSink out;

    __device__ __forceinline__ void init(uint8_t *stripe, uint32_t cap) { x1 = 0; x2 = 0xFFFFFFFFu; rev = 0; out.init(stripe, cap); }

    __device__ __forceinline__ void emit(uint32_t v, uint32_t n) {  // n in 1..32, v = top n bits of x1
        if (rev == 0u) { out.put(v, n); return; }
        const uint32_t b = (v >> (n - 1u)) & 1u;
        const uint32_t fill = b ? 0u : 0xFFFFFFFFu;
        if (n + rev <= 32u) {
            uint32_t rest = v & ((1u << (n - 1u)) - 1u);
            uint32_t mid = (fill & ((1u << rev) - 1u)) << (n - 1u);
            out.put((b << (n + rev - 1u)) | mid | rest, n + rev);
        } else {
            out.put(b, 1u);
            uint32_t r = rev;
            while (r > 0u) {
                uint32_t k = r < 32u ? r : 32u;
                out.put(k == 32u ? fill : (fill & ((1u << k) - 1u)), k);
                r -= k;
            }
            if (n > 1u) out.put(v & ((1u << (n - 1u)) - 1u), n - 1u);
        }
        rev = 0u;
    }

    __device__ __forceinline__ void encode(uint32_t bit, uint32_t prob) {
        const uint32_t p32 = prob ? (prob << 16) : 1u;              // lerp :111
        const uint32_t xmid = x1 + __umulhi(x2 - x1, p32);          // :112-116
        if (bit) x2 = xmid; else x1 = xmid + 1u;                    // :45-48
        const uint32_t n = (uint32_t)__clz((int)(x1 ^ x2));         // 32 when x1 == x2
        if (n) {
            if (n == 32u) { emit(x1, 32u); x1 = 0u; x2 = 0xFFFFFFFFu; }
            else { emit(x1 >> (32u - n), n); x1 <<= n; x2 = (x2 << n) | ((1u << n) - 1u); }
        }
        const uint32_t m = (uint32_t)__clz((int)~((x1 & ~x2) << 1)); // 0..31
        x1 = (x1 << m) & 0x7FFFFFFFu;
        x2 = (x2 << m) | 0x80000000u | ((1u << m) - 1u);
        rev += m;
    }

    // ACStats (helpers.rs:60-90), the counting sink: bits written so far (write_bit adds 1 + rev_bits; pending parity bits not
    // yet resolved do not count; flush adds nothing, :87-89).  Valid before flush().
    __device__ __forceinline__ uint32_t stats_bits() const { return 8u * out.pos + out.nb; }

    // ArithmeticCoder::flush -> ACWriter::flush(x2)  (arithmetic_coder.rs:67-71, io.rs:91-100)
    __device__ __forceinline__ uint32_t flush() {
        emit(x2 >> 31, 1u);
        const uint32_t idx = out.nb & 7u;
        if (idx) {
            const uint32_t k = 8u - idx;
            out.put((x2 << 1) >> (32u - k), k);
        }
        out.finish();
        return out.pos;
    }
};

// ---------------------------------------------------------------------------
// ACReader + decoder side (io.rs:7-49, arithmetic_coder.rs:74-106)
// ---------------------------------------------------------------------------
struct BitSource {
    // ACReader (entropy_coding/io.rs:7-49): MSB-first bits, zeros past EOF (:23-26).  The stream is read four bytes at a
    // time and ONE WORD AHEAD of its use (a byte-at-a-time read on demand put a dependent global load — ~1 us for a
    // lone lane — into every few bit-steps of the decoder).
    const uint8_t *p; uint32_t len, pos; uint64_t win; uint32_t navail, nxt;
    __device__ __forceinline__ uint32_t load_word(uint32_t at) const {   // bytes [at, at+4) big-endian, zeros past len
        uint32_t v = 0u;
        if (at + 4u <= len) { uint32_t raw; __builtin_memcpy(&raw, p + at, 4); v = __builtin_bswap32(raw); }
        else { for (uint32_t k = 0; k < 4u; k++) v = (v << 8) | (at + k < len ? p[at + k] : 0u); }
        return v;
    }
    __device__ __forceinline__ void init(const uint8_t *s, uint32_t l) { p = s; len = l; pos = 4u; win = 0; navail = 0; nxt = load_word(0u); }
    __device__ __forceinline__ uint32_t get(uint32_t k) {  // k in 0..32
        if (navail < k) {            // refill: at most 31 bits are pending, so 32 more fit
            win = (win << 32) | nxt;
            navail += 32u;
            nxt = load_word(pos);    // consumed one refill later
            pos += 4u;
        }
        uint32_t v = (uint32_t)(win >> (navail - k));
        if (k < 32u) v &= (1u << k) - 1u;
        navail -= k;
        return v;
    }
};

struct Decoder {
    uint32_t x1, x2, x;
    BitSource in;
    __device__ __forceinline__ void init(const uint8_t *s, uint32_t l) { in.init(s, l); x1 = 0; x2 = 0xFFFFFFFFu; x = in.get(32u); }
    // Model probabilities are never 0 (Counter::p, the state table and the APM clamp all give 1..65535): with range >= 2^30
    // before the step the new range is >= 2^14 - 1, so x1 != x2 afterwards and both renormalisation loops collapse into ONE
    // shift by s = n + m <= 18 found with a single clz (as in the encoder, w3_coder.h): n equal leading bits, then the m E3
    // positions where x1 = 1 and x2 = 0.  Of the E3 loop's XORs on x only the last survives the shifts, and there was an
    // E3 step exactly when the raw shifted x1 has its top bit set.
    __device__ __forceinline__ uint32_t decode(uint32_t prob) {
        if (__builtin_expect(prob == 0u, 0)) return decode_general(prob);
        const uint32_t xmid = x1 + __umulhi(x2 - x1, prob << 16);
        const uint32_t bit = x <= xmid;                                // :82
        if (bit) x2 = xmid; else x1 = xmid + 1u;
        const uint32_t s = (uint32_t)__builtin_clz((x1 ^ x2) & (((~x1 | x2) << 1) | 1u));
        const uint32_t X1 = x1 << s, X2 = ((x2 + 1u) << s) - 1u;
        x = ((x << s) | in.get(s)) ^ (X1 & 0x80000000u);
        x1 = X1 & 0x7FFFFFFFu;
        x2 = X2 | 0x80000000u;
        return bit;
    }
    // the same for a caller whose probabilities are never 0 (no fallback inlined beside every step)
    __device__ __forceinline__ uint32_t decode_nz(uint32_t prob) {
        const uint32_t xmid = x1 + __umulhi(x2 - x1, prob << 16);
        const uint32_t bit = x <= xmid;
        if (bit) x2 = xmid; else x1 = xmid + 1u;
        const uint32_t s = (uint32_t)__builtin_clz((x1 ^ x2) & (((~x1 | x2) << 1) | 1u));
        const uint32_t X1 = x1 << s, X2 = ((x2 + 1u) << s) - 1u;
        x = ((x << s) | in.get(s)) ^ (X1 & 0x80000000u);
        x1 = X1 & 0x7FFFFFFFu;
        x2 = X2 | 0x80000000u;
        return bit;
    }
    __device__ __forceinline__ uint32_t decode_general(uint32_t prob) {
        const uint32_t p32 = prob ? (prob << 16) : 1u;
        const uint32_t xmid = x1 + __umulhi(x2 - x1, p32);
        const uint32_t bit = x <= xmid;                                // :82
        if (bit) x2 = xmid; else x1 = xmid + 1u;
        const uint32_t n = (uint32_t)__clz((int)(x1 ^ x2));
        if (n == 32u) { x = in.get(32u); x1 = 0u; x2 = 0xFFFFFFFFu; }
        else if (n) { x = (x << n) | in.get(n); x1 <<= n; x2 = (x2 << n) | ((1u << n) - 1u); }
        const uint32_t m = (uint32_t)__clz((int)~((x1 & ~x2) << 1));
        if (m) {
            // m iterations of x = ((x<<1) ^ Q2) | bit : only the last XOR survives the shifts
            x = ((x << m) | in.get(m)) ^ 0x80000000u;
            x1 = (x1 << m) & 0x7FFFFFFFu;
            x2 = (x2 << m) | 0x80000000u | ((1u << m) - 1u);
        }
        return bit;
    }
};

// ---------------------------------------------------------------------------
// ACHistory::hash (history/ac_history.rs:28-46) with EntropyWriter (:50-89).
// The writer shifts bits in from the top and the result is state>>(32-idx),
// i.e. the k-th written bit lands at bit k: accumulate LSB-first.  An Err from
// the writer (idx == max_bits) ends the loop; max_bits == 0 or idx == 0 -> 0.
// `bits` = last 64 input bits (newest at bit 0), `pos` = bits seen so far.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ac_history_hash(uint64_t bits, uint32_t pos, uint32_t max_bits, const uint16_t *table) {
    uint32_t x1 = 0u, x2 = 0xFFFFFFFFu, hash = 0u, idx = 0u, rev = 0u;
    uint32_t al = pos & 7u;                                            // model.align(pos & 7)  :36
    for (int i = 0; i < 64; i++) {
        const uint32_t bit = (uint32_t)(bits >> i) & 1u;               // most recent bit first :38
        al = (al + 7u) & 7u;                                           // stationary.rs:54-57
        const uint32_t prob = table[al];
        const uint32_t p32 = prob ? (prob << 16) : 1u;
        const uint32_t xmid = x1 + __umulhi(x2 - x1, p32);
        if (bit) x2 = xmid; else x1 = xmid + 1u;
        while (((x1 ^ x2) >> 31) == 0u) {
            const uint32_t b = x1 >> 31;
            if (idx == max_bits) return hash;
            hash |= b << idx; idx++;
            while (rev > 0u) {
                rev--;
                if (idx == max_bits) return hash;
                hash |= (b ^ 1u) << idx; idx++;
            }
            x1 <<= 1; x2 = (x2 << 1) | 1u;
        }
        while (x1 >= 0x40000000u && x2 < 0xC0000000u) {
            rev++;
            x1 = (x1 << 1) & 0x7FFFFFFFu;
            x2 = (x2 << 1) | 0x80000001u;
        }
    }
    return hash;
}

// The same hash without data-dependent inner loops (k_achash computes it for every step of every block: the literal
// form above spends most of its time in divergent one-bit-at-a-time loops).  Per history bit: the n equal leading bits
// of (x1, x2) are written at once — brev(x1) holds them in write order, the pending parity bits (rev copies of the
// complement, io.rs:84-87 semantics in EntropyWriter::write_bit, ac_history.rs:63-83) go right after the first one —
// and the E3 run length is one more clz.  A lane is done once max_bits bits are written (the writer's Err).
// `rot[r]` = the lerp operand (prob << 16, or 1 for prob 0: arithmetic_coder.rs:111) of the r-th coded history bit and
// of every 8th after it: StationaryModel::predict walks the bit positions backwards from `pos & 7` (stationary.rs:54-57),
// so the caller rotates the 8-entry table once per bit position and no step looks anything up.
// State of the nested coder after some history bits (also the entry type of the 8-bit prefix table of k_achash).
struct ACHashState {
    uint32_t x1, x2, hash, meta;   // meta = idx | rev << 8 | done << 31   (idx <= 32 + 31 + 8*31, rev <= 8*31 inside a table entry)
};
__device__ __forceinline__ ACHashState ac_hash_state_init(uint32_t max_bits) {
    ACHashState s; s.x1 = 0u; s.x2 = 0xFFFFFFFFu; s.hash = 0u; s.meta = max_bits == 0u ? 0x80000000u : 0u;
    return s;
}

// Steps [i_begin, i_end) (multiples of 8) of the hash from state `st`; returns the state after them.
__device__ __forceinline__ ACHashState ac_history_hash_steps(uint64_t bits, uint32_t max_bits, const uint32_t (&rot)[8], ACHashState st,
                                                            int i_begin, int i_end) {
    uint32_t x1 = st.x1, x2 = st.x2, hash = st.hash, idx = st.meta & 0xFFu, rev = (st.meta >> 8) & 0x7FFFFFu;
    bool done = (st.meta >> 31) != 0u;
    for (int i0 = i_begin; i0 < i_end; i0 += 8) {
        if (!__ballot(!done)) break;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            // (every second step: a wavefront whose lanes all have their max_bits bits stops here, not at the end of the group of eight —
            //  with small max_bits most lanes need one or two history bits beyond what the prefix table covered)
            if ((r == 2 || r == 4 || r == 6) && !__ballot(!done)) goto all_done;
            if (!done) {
                const uint32_t bit = (uint32_t)(bits >> (i0 + r)) & 1u;
                const uint32_t xmid = x1 + __umulhi(x2 - x1, rot[r]);
                if (bit) x2 = xmid; else x1 = xmid + 1u;
                const uint32_t a = x1 ^ x2;
                const uint32_t n = a ? (uint32_t)__builtin_clz(a) : 32u;
                if (n) {
                    const uint32_t rb = __builtin_bitreverse32(x1) & (n == 32u ? 0xFFFFFFFFu : ((1u << n) - 1u));   // the n bits, first written at bit 0
                    const uint32_t rr = rev < 40u ? rev : 40u;                                                      // (bits past 32 never matter)
                    const uint64_t par = (rb & 1u) ? 0ull : ((1ull << rr) - 1ull);                                  // rr copies of the complement
                    const uint64_t seq = (uint64_t)(rb & 1u) | (par << 1) | ((uint64_t)(rb >> 1) << (1u + rr));
                    hash |= (uint32_t)(seq << idx);
                    idx += n + rev;
                    rev = 0u;
                    x1 = n == 32u ? 0u : x1 << n;
                    x2 = n == 32u ? 0xFFFFFFFFu : ((x2 << n) | ((1u << n) - 1u));
                }
                if (idx >= max_bits) done = true;
                // E3 run: positions below the top bit where x1 has 1 and x2 has 0 (arithmetic_coder.rs:57-62)
                const uint32_t e3 = (x1 & ~x2) << 1;
                const uint32_t m = (~e3) ? (uint32_t)__builtin_clz(~e3) : 32u;   // <= 31: bit 0 of e3 is 0
                rev += m;
                x1 = (x1 << m) & 0x7FFFFFFFu;
                x2 = (x2 << m) | 0x80000000u | ((1u << m) - 1u);
            }
        }
    }
all_done:
    ACHashState o;
    o.x1 = x1; o.x2 = x2; o.hash = hash;
    o.meta = (idx < 255u ? idx : 255u) | (rev << 8) | (done ? 0x80000000u : 0u);   // idx >= max_bits (<= 32) only ever means "done"
    return o;
}
__device__ __forceinline__ uint32_t ac_hash_finish(const ACHashState &s, uint32_t max_bits) {
    return max_bits >= 32u ? s.hash : (s.hash & ((1u << max_bits) - 1u));
}
__device__ __forceinline__ uint32_t ac_history_hash_fast(uint64_t bits, uint32_t max_bits, const uint32_t (&rot)[8]) {
    return ac_hash_finish(ac_history_hash_steps(bits, max_bits, rot, ac_hash_state_init(max_bits), 0, 64), max_bits);
}

}  // namespace w3
