// w3_predict_wave.h — PREDICT for every Counter-table leaf the sorted kernels of w3_predict.h do not cover: any
// bits_in_context / alignment_bits (models/ordern.rs:26-44: OrderN(22,2), OrderN(30,1), OrderN(32,1) — the reference's best
// plain configuration, bin/ordern/enwik7.log:163) and hashed histories wider than 8 bits (models/ordern_entropy.rs:27-46 with
// ACHistory / HuffHistory / RawHistory: (26,3)+ACHistory(23) and (20,3)+ACHistory(17), the reference's best ratios,
// bin/entropy-hashing-ac/main.rs:21-25).  Before round 3 these ran on the lane-per-block kernel (one serial chain of 8 * block_size
// dependent HBM round trips per block: 0.1 - 0.3 GiB/s).
//
// k_predict_wave: one WAVEFRONT per block, time order, 64 consecutive bit-steps per round (lane = step), the leaf's Counter table
// in HBM, private to the wavefront for the block's life:
//   * ctx of every step from the input bits alone (raw history: an unaligned 8-byte window per lane) or from the 32-bit hashes a
//     key kernel wrote (k_achash32 / k_huffkeys32);
//   * table = direct-indexed [2^bits] u32 when that is no larger than the exact map, else an EXACT open-addressing map of
//     2 * 8 * block_size slots {key, n0 | n1 << 16} (the reference's tables are direct-indexed and collision free: a lossy hash
//     would break parity); slots are claimed with a 32-bit compare-and-swap, so two lanes of a round that meet one empty slot
//     settle it; ctx 0 has a slot of its own (0 = empty);
//   * lanes of a round that share a context are found with ballots over the SLOT index (a 6-bit pre-test first: wide contexts
//     rarely repeat inside 64 steps) and combined as in rank_round (w3_predict.h): the first lane's Counter + the lanes below,
//     the last lane writes back; Counter::update's halving (counter.rs:22-25) by serial replay of the context that reaches it;
//   * the round's 64 probabilities leave as ONE coalesced 128-byte store (time order: no scatter).
// Counts are read and written with agent-scope accesses (they bypass the L1, which a store of another lane does not update),
// and a round's stores are waited for before the next round's loads are issued.
#pragma once
#include "w3_predict.h"

namespace w3 {

struct WaveArgs {
    const uint8_t *in;
    uint64_t n;
    uint32_t block_size, nblocks;
    uint16_t *P;             // [8 n] this leaf's stream (u16 per bit-step, time order)
    const uint32_t *keys32;  // [8 n] History::hash() of every step (k_achash32 / k_huffkeys<true>), or null: raw history from the input
    uint8_t *tables;         // one table per resident wavefront
    uint64_t table_stride;
    uint32_t align;          // alignment_bits
    uint32_t hist_mask;      // 2^(bits - align) - 1
    uint32_t use_hash;       // 0: direct-indexed u32 [2^bits]; 1: exact map
    uint32_t hash_slots;     // power of two (use_hash)
};

__device__ __forceinline__ uint32_t wv_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wv_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// lanes whose `v` (nbits wide) equals mine
template <int NBITS>
__device__ __forceinline__ uint64_t match_value(uint32_t v) {
    uint64_t m = ~0ull;
#pragma unroll
    for (int k = 0; k < NBITS; k++) {
        const bool mybit = (v >> k) & 1u;
        const uint64_t B = __ballot(mybit);
        m &= mybit ? B : ~B;
    }
    return m;
}

__device__ __forceinline__ uint64_t wv_load64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool KEYS>
__global__ void __launch_bounds__(64) k_predict_wave(WaveArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t gt = lane_gt_mask();
    uint8_t *tbl8 = a.tables + (uint64_t)blockIdx.x * a.table_stride;
    uint32_t *tbl = reinterpret_cast<uint32_t *>(tbl8);
    uint64_t *tbl64 = reinterpret_cast<uint64_t *>(tbl8);
    const uint32_t amask = (1u << a.align) - 1u;
    const uint64_t tbl_words = a.use_hash ? 2ull * a.hash_slots + 2ull : (uint64_t)a.table_stride / 4u;
    for (uint32_t b = blockIdx.x; b < a.nblocks; b += gridDim.x) {
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
        const uint8_t *blk = a.in + off;
        const uint32_t first = window_head(off, 7u);
        // a fresh model: every Counter (0, 0), the map empty
        for (uint64_t w = (uint64_t)lane * 4u; w < tbl_words; w += 256u) *reinterpret_cast<uint4 *>(tbl + w) = make_uint4(0u, 0u, 0u, 0u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t nsteps = len * 8u, last_step = nsteps - 1u;
        // a round's operands (input window, hash) do not depend on the table: they are requested one round ahead
        uint64_t Wn = wave_window(blk, min(lane, last_step) >> 3, first);
        uint32_t Kn = 0u;
        if constexpr (KEYS) Kn = a.keys32[off * 8u + min(lane, last_step)];
        for (uint32_t t0 = 0; t0 < nsteps; t0 += 64u) {
            const uint32_t t = t0 + lane, j = t & 7u;
            const bool valid = t < nsteps;
            const uint64_t W = Wn;
            const uint32_t K = Kn;
            {
                const uint32_t tn = min(t + 64u, last_step);
                Wn = wave_window(blk, tn >> 3, first);
                if constexpr (KEYS) Kn = a.keys32[off * 8u + tn];
            }
            const uint32_t bit = (uint32_t)(W >> (7u - j)) & 1u;
            uint32_t ctx;
            if constexpr (KEYS) ctx = t == 0u ? 0u : ((K & a.hist_mask) << a.align) | (t & amask);
            else ctx = (((uint32_t)(W >> (8u - j))) & a.hist_mask) << a.align | (t & amask);   // (t == 0: history and alignment are 0)
            // the Counter's slot and its counts in ONE access: {key, n0 | n1 << 16} (exact map) or the u32 itself (direct)
            uint32_t slot, base = 0u;   // slot: index of the counts word in tbl
            if (!a.use_hash) { slot = ctx; if (valid) base = wv_load(&tbl[slot]); }
            else if (ctx == 0u) { slot = 2u * a.hash_slots + 1u; if (valid) base = wv_load(&tbl[slot]); }
            else {
                uint32_t h = (ctx * 2654435761u) ^ (ctx >> 15);
                slot = 0u;
                bool found = !valid;
                while (!found) {   // (every probe sequence ends: the map has twice as many slots as a block has steps)
                    h &= a.hash_slots - 1u;
                    const uint64_t kv = wv_load64(&tbl64[h]);
                    uint32_t k = (uint32_t)kv, c = (uint32_t)(kv >> 32);
                    if (k == 0u) {   // empty: claim it (a lane of this round with another context may get there first)
                        uint32_t expect = 0u;
                        __hip_atomic_compare_exchange_strong(&tbl[2u * h], &expect, ctx, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        k = expect == 0u ? ctx : expect;
                        c = 0u;   // (nobody has written counts to a slot claimed in this round: stores come at the round's end)
                    }
                    if (k == ctx) { slot = 2u * h + 1u; base = c; found = true; }
                    h++;
                }
            }
            // lanes of this round on the same Counter (they all read the same counts)
            uint64_t M = match_value<6>(slot) & __ballot(valid);
            if (__ballot(valid && (M & (M - 1ull)) != 0ull)) M = match_value<30>(slot >> (a.use_hash ? 1 : 0)) & match_value<2>(slot >> 30) & __ballot(valid);
            const uint64_t ones = __ballot(bit != 0u) & M;
            const uint32_t n1l = mbcnt64(ones), n0l = mbcnt64(M) - n1l;
            uint32_t s0 = (base & 0xFFFFu) + n0l, s1 = (base >> 16) + n1l;
            const bool last = valid && (M & gt) == 0ull;
            const uint32_t f0 = s0 + (bit ^ 1u), f1 = s1 + bit;
            uint32_t f = f0 | (f1 << 16);
            // Counter::update halves both counts when one reaches 65535 (counter.rs:22-25): replay such a context serially
            uint64_t satm = __ballot(last && (f0 >= 65535u || f1 >= 65535u));
            while (satm) {
                const int k = __ffsll((long long)satm) - 1;
                const uint64_t Mc = readlane_u64(M, k), Oc = readlane_u64(ones, k);
                uint32_t st = readlane_u32(base, k);
                uint64_t it = Mc;
                while (it) {
                    const int m = __ffsll((long long)it) - 1;
                    it &= it - 1;
                    if ((int)lane == m) { s0 = st & 0xFFFFu; s1 = st >> 16; }
                    st = counter_update_packed(st, (uint32_t)(Oc >> m) & 1u);
                }
                if ((int)lane == k) f = st;
                satm &= satm - 1;
            }
            const uint32_t p = counter_p(s0, s1);
            if (last) wv_store(&tbl[slot], f);
            if (valid) a.P[off * 8u + t] = (uint16_t)p;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the round's stores are on their way before the next round's loads
            __builtin_amdgcn_s_waitcnt(0);
        }
    }
}

// ---------------------------------------------------------------------------
// 32-bit context hashes of every bit-step for the wide OrderNEntropy leaves (HuffHistory: k_huffkeys<true>, w3_predict.h).
// k_achash32: ACHistory::hash (history/ac_history.rs:28-46) per (byte position, bit position): the coder state after the 16 most
// recent history bits comes from k_achash_lut's table (w3_predict.h), the rest is coded from there.  One thread per step.
// ---------------------------------------------------------------------------
struct Hash32Args {
    const uint8_t *in; uint64_t n; uint32_t block_size; uint32_t max_bits; uint16_t table[8];
    const uint4 *lut;          // [65536][8] (k_achash_lut)
    uint32_t *keys32;          // [8 n]
};

__global__ void __launch_bounds__(256) k_achash32(Hash32Args a) {
    // step index over the whole input: g = 8 * byte + j.  A grid-stride loop: an AQL dispatch counts its work-items in 32 bits, so a launch
    // of one thread per step would silently cover only 8 n mod 2^32 steps of an input of 2^29 bytes or more (it did until round 4:
    // tests/test_gpu_parity.py::test_keyed_leaf_beyond_2_pow_32_steps).
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < a.n * 8u; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t pos = g >> 3;
        const uint32_t j = (uint32_t)g & 7u;
        const uint64_t b = pos / a.block_size;
        const uint32_t i = (uint32_t)(pos - b * a.block_size);
        const uint8_t *blk = a.in + b * a.block_size;
        // the last 64 bits before bit j of byte i, newest at bit 0 (zeros before the block start)
        uint64_t hist = 0;
        for (uint32_t k = 1; k <= 8 && k <= i; k++) hist |= (uint64_t)blk[i - k] << (8 * (k - 1));
        hist = (hist << j) | (uint64_t)(blk[i] >> (8u - j));
        uint32_t p32t[8], rot[8];
#pragma unroll
        for (int k = 0; k < 8; k++) p32t[k] = a.table[k] ? ((uint32_t)a.table[k] << 16) : 1u;   // lerp operand, arithmetic_coder.rs:111
        // StationaryModel::predict walks the bit positions backwards from j: the r-th coded history bit uses table[(j - 1 - r) & 7]
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t k = (j + 7u - (uint32_t)r) & 7u;
            rot[r] = k == 0 ? p32t[0] : k == 1 ? p32t[1] : k == 2 ? p32t[2] : k == 3 ? p32t[3] : k == 4 ? p32t[4] : k == 5 ? p32t[5] : k == 6 ? p32t[6] : p32t[7];
        }
        const uint4 sv = a.lut[((uint32_t)hist & ((1u << W3_ACHASH_LUT_BITS) - 1u)) * 8u + j];
        ACHashState st; st.x1 = sv.x; st.x2 = sv.y; st.hash = sv.z; st.meta = sv.w;
        st = ac_history_hash_steps(hist, a.max_bits, rot, st, W3_ACHASH_LUT_BITS, 64);
        a.keys32[g] = ac_hash_finish(st, a.max_bits);
    }
}

}  // namespace w3
