// w3_predict.h — PREDICT phase of the two-phase encoder (gfx950).
//
// The encoder knows every future context: a Counter-table model's context at
// step t is a function of the input bits only (models/ordern.rs:35-43), and
// the Counter state it will find there is the running count of the bits seen
// earlier in that context (models/counter.rs:20-26).  So Model::predict for
// ALL steps of a block is a "rank among equal keys" problem, solved here by one
// wavefront per block with ballot-match ranking, 64 byte positions per round
// and all 8 bit positions of a byte per lane:
//
//   key_j(i)  = last H bits before bit j of byte i      (alignment_bits == 3)
//   state     = LDS table[j][key]  (+ the lanes below me with the same key)
//   p_j(i)    = Counter::p(state)
//
// H <= 8  : one table, positions in time order                 (Order0, OrderN(<=11,3),
//           OrderNEntropy(<=11,3,h) with keys from k_achash)
// H == 16 : positions stably partitioned by the previous byte c1, then ranked by
// H == 24 : the remaining 8 key bits inside each (c1[,c2]) group (Order1, OrderN(27,3)).
//           The table describes the one group still open at a round boundary; groups that
//           fit inside a round never touch it.
//
// Counter saturation (the halve-both rule at 65535) is handled exactly: the
// table holds the true Counter state, and a context that would saturate inside
// a round is replayed serially with scalar code (rare).
//
// Every leaf writes its own u16 stream (time-ordered kernels coalesced; the
// partitioned kernels scatter 16 B per position, write-only: a read-modify-write
// of a shared stream doubled the HBM traffic of those kernels).  k_mix then merges
// the streams into ONE stream P with OpinionMixer2's rule (leftmost leaf of
// maximal |p-1/2|; see w3_device.h), so the coder reads 16 B per input byte.
#pragma once
#include "w3_device.h"
#include "../../include/w3hip.h"

namespace w3 {

#define W3_PF 8   // rounds whose loads are in flight together (one batch)

// "Half-CU" form of the predict kernels (template parameter NW > 1): NW independent wavefronts share ONE workgroup whose
// static LDS is padded to W3_HALF_CU_LDS = 80 KiB, half of a CU's 160 KiB: one such workgroup and ONE workgroup of the previous
// call's APM stage (k_apm0: 79,968 B) or coder (k_coder_x5: 74,272 B) share a CU without fragmenting its LDS (DESIGN.md section
// 2.8).  Measured (profiles/r3_pipeline/): a workgroup of MORE than 80 KiB (82,000 B was tried first) is never placed beside
// another large one — it waits for an empty CU — so nothing here exceeds the half.  The wavefronts of a workgroup never
// synchronise with each other (wave barriers only).  NW == 1 is the plain one-wavefront workgroup.
#define W3_HALF_CU_LDS 81920u
template <int NW, size_t USED> struct HalfCuPad { static constexpr size_t words = (NW > 1 && USED < W3_HALF_CU_LDS) ? (W3_HALF_CU_LDS - USED + 3u) / 4u : (NW > 1 ? 0u : 1u); };
template <size_t WORDS> struct HalfCuPadDecl {
    static __device__ __forceinline__ void keep(uint64_t n) {
        __shared__ uint32_t pad_[WORDS];
        if (n == ~0ull) pad_[threadIdx.x % WORDS] = 1u;   /* never true: keeps the padding allocated */
    }
};
template <> struct HalfCuPadDecl<0> { static __device__ __forceinline__ void keep(uint64_t) {} };   // (the wavefronts fill the half exactly)
#define W3_HALF_CU_PAD(NW, USED) HalfCuPadDecl<((NW) > 1 ? HalfCuPad<NW, (USED)>::words : 0u)>::keep(a.n);

struct PredictArgs {
    const uint8_t *in;      // original bytes (device)
    uint64_t n;
    uint32_t block_size, nblocks;
    uint4 *P;               // [n] this leaf's stream: 8 x u16 per input byte, block-major (same index as `in`)
    const uint2 *keys;      // [n] 8 x u8 precomputed keys per byte (k_achash) or null
    uint32_t *perm;         // partition: per-wave scratch, 2 * block_size records
    uint2 *rec;             // [n] sorted records (position, window bytes) of every block
    uint4 *sink;            // 64 x 16 B nobody reads: stores of predicated-off lanes go here, so that every round issues the same
                            // memory operations and hipcc can count them (a branch around a store costs s_waitcnt vmcnt(0) per round)
    const uint2 *rec_src;   // k_partition<3>: [n] records already sorted by c1 (an Order1 leaf's a.rec), else null
    uint32_t *splits;       // [nblocks][W3_SLICES + 1] slice boundaries inside each block's sorted range
    uint32_t *job_counter;  // k_rank_sorted: next job (zeroed before the launch)
    uint32_t hbits;         // H = bits_in_context - 3
    uint32_t maxseg;        // k_rank_sorted: rounds with more groups than this take the ballot path (W3_ATOMIC_MAXSEG)
    uint32_t dbg_flags;      // bit0 = skip the stream stores (timing experiments only); bit1 = ballot rounds only (no LDS atomics);
                             // bit3 = FAULT INJECTION for the tests of the sampled verification: one returning add of every block hands two lanes each other's value
    unsigned long long *dbg; // optional: per-phase s_memtime sums (diagnostic builds/runs only; never read by kernels)
    uint32_t fault_block;    // dbg_flags bit3: the one block the injected fault hits, or 0xFFFFFFFF = every block
};

#define W3_STAMP(slot)                                                                         \
    do {                                                                                       \
        if (a.dbg) {                                                                           \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                        \
            if ((threadIdx.x & 63) == 0) atomicAdd(&a.dbg[slot], t_ - t_prev);                 \
            t_prev = t_;                                                                       \
        }                                                                                      \
    } while (0)

__device__ __forceinline__ uint64_t lane_lt_mask() { return (1ull << (threadIdx.x & 63)) - 1ull; }
__device__ __forceinline__ uint64_t lane_gt_mask() { return ~((2ull << (threadIdx.x & 63)) - 1ull); }

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, k); }
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int k) {
    uint32_t lo = readlane_u32((uint32_t)v, k), hi = readlane_u32((uint32_t)(v >> 32), k);
    return ((uint64_t)hi << 32) | lo;
}

// Match masks of the 8 sliding H-bit windows of w16 (window j = bits [8-j, 8-j+H) ).
template <int H>
__device__ __forceinline__ void match_windows(uint32_t w16, uint64_t M[8]) {
    uint64_t X[16];
#pragma unroll
    for (int b = 1; b < 16; b++) {
        if (b >= 1 && b <= 7 + H) {
            const bool mybit = (w16 >> b) & 1u;
            const uint64_t B = __ballot(mybit);
            X[b] = mybit ? B : ~B;
        }
    }
    if constexpr (H == 8) {
        uint64_t A2[15], A4[13];
#pragma unroll
        for (int b = 1; b <= 14; b++) A2[b] = X[b] & X[b + 1];
#pragma unroll
        for (int b = 1; b <= 12; b++) A4[b] = A2[b] & A2[b + 2];
#pragma unroll
        for (int j = 0; j < 8; j++) M[j] = A4[8 - j] & A4[12 - j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t m = ~0ull;
#pragma unroll
            for (int k = 0; k < H; k++) m &= X[8 - j + k];
            M[j] = m;
        }
    }
}

// Match masks when the 8 keys are given explicitly (one byte per bit position).
template <int H>
__device__ __forceinline__ void match_keys(uint2 k8, uint64_t M[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t key = ((j < 4 ? k8.x : k8.y) >> (8 * (j & 3))) & 0xFFu;
        // lanes whose key bit k equals mine: mybit ? B : ~B  ==  ~(B ^ sign-extended bit); folded into the running AND as one
        // 3-input bit operation per 32-bit half (v_bitop3_b32) instead of two selects and two ANDs
        uint32_t mlo = 0xFFFFFFFFu, mhi = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < H; k++) {
            const uint32_t sm = (uint32_t)__builtin_amdgcn_sbfe((int)key, k, 1);
            const uint64_t B = __ballot(sm != 0u);
            mlo &= ~((uint32_t)B ^ sm);
            mhi &= ~((uint32_t)(B >> 32) ^ sm);
        }
        M[j] = ((uint64_t)mhi << 32) | mlo;
    }
}

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {  // bits of m below my lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// One round: 64 positions x 8 bit positions.  Returns the 8 probabilities.
//   c0      : the byte being coded at this position (bit j = (c0 >> (7-j)) & 1)
//   key[j]  : table index per bit position
//   M[j]    : lanes with the same key_j (any group)
//   seg     : lanes that are valid and in the same group as me (all valid lanes when not GROUPED)
//   rd      : GROUPED: this lane's group is the one the table currently describes (else its Counters are new)
// The table holds the exact Counter state of every context of the current group.
// !GROUPED: the last lane of a context writes the state back at once.
// GROUPED : the write-back is left to the caller (it may have to clear the table first):
//           fin[j] = state after the round, wmask bit j = this lane is the context's last lane.
template <bool GROUPED>
__device__ __forceinline__ void rank_round(uint32_t c0, const uint32_t key[8], const uint64_t M[8], uint64_t seg, bool valid, bool rd,
                                           uint32_t *tbl, uint32_t p[8], uint32_t fin[8], uint32_t &wmask) {
    // Lanes communicate through `tbl` (one lane writes a context's state, others read it a round later).  To the
    // compiler that is a data race: with constant indices (H == 0) it forwarded each lane's own stale value
    // instead of re-reading LDS.  A compiler-level memory barrier per round makes it re-load (volatile accesses
    // also work but serialise the eight reads: +25 % kernel time).
    __asm__ volatile("" ::: "memory");
    const uint64_t gt = lane_gt_mask();
    const int lane = threadIdx.x & 63;
    wmask = 0u;
    // the eight table reads are independent (one table per bit position): issue them together
    uint32_t basev[8];
#pragma unroll
    for (int j = 0; j < 8; j++) basev[j] = tbl[j * 256 + key[j]];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint64_t Mj = M[j] & seg;
        const uint32_t bit = (c0 >> (7 - j)) & 1u;
        const uint64_t ones = __ballot(bit) & Mj;
        uint32_t base = basev[j];
        if constexpr (GROUPED) base = rd ? base : 0u;
        const uint32_t b0 = base & 0xFFFFu, b1 = base >> 16;
        const uint32_t n1l = mbcnt64(ones), n0l = mbcnt64(Mj) - n1l;   // same-context lanes below me, by coded bit
        uint32_t s0 = b0 + n0l, s1 = b1 + n1l;                          // Counter state this lane predicts from
        const bool last = valid && (Mj & gt) == 0ull;                   // last lane of its context in this round
        uint32_t f0 = s0 + (bit ^ 1u), f1 = s1 + bit;                   // state after my own update (meaningful on `last`)
        // Counter::update halves both counts when one reaches 65535 (counter.rs:22-25):
        // replay such a context serially (uniform scalar loop; rare)
        uint64_t satm = __ballot(last && (f0 >= 65535u || f1 >= 65535u));
        uint32_t f = f0 | (f1 << 16);
        while (satm) {
            const int k = __ffsll((long long)satm) - 1;
            const uint64_t Mc = readlane_u64(Mj, k);
            const uint64_t Oc = readlane_u64(ones, k);
            uint32_t st = readlane_u32(base, k);
            uint64_t it = Mc;
            while (it) {
                const int m = __ffsll((long long)it) - 1;
                it &= it - 1;
                if (lane == m) { s0 = st & 0xFFFFu; s1 = st >> 16; }
                st = counter_update_packed(st, (uint32_t)(Oc >> m) & 1u);
            }
            if (lane == k) f = st;
            satm &= satm - 1;
        }
        p[j] = counter_p(s0, s1);
        fin[j] = f;
        wmask |= last ? (1u << j) : 0u;
    }
    if constexpr (!GROUPED) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if ((wmask >> j) & 1u) tbl[j * 256 + key[j]] = fin[j];
        __asm__ volatile("" ::: "memory");
    }
}

// ---------------------------------------------------------------------------
// The same round with LDS atomics: ds_add_rtn_u32 serialises the lanes of a wavefront that hit one address in ascending
// lane order (not in the ISA manual: measured, tools/lds_atomic_order.hip, and checked per device at context creation —
// k_lds_order_selftest; without it the ballot rounds above run), so ONE returning add per bit position hands every lane
// the packed Counter (n0 | n1 << 16) exactly as its position sees it, time order = lane order, and leaves the table
// updated: no match masks, no ranks, no write-back.  What it cannot do is Counter::update's halving at 65535
// (counter.rs:22-25): a wavefront whose round returns a count >= W3_ATOMIC_LIM drops to the ballot rounds for the rest
// of the table's life.  Safe: a count grows by at most 64 per round, so a round that starts below LIM + 64 ends below 65535.
// ---------------------------------------------------------------------------
#define W3_ATOMIC_LIM 65400u
#ifndef W3_ATOMIC_MAXSEG
#define W3_ATOMIC_MAXSEG 4u   // k_rank_sorted: rounds with more groups than this take the ballot path
#endif
typedef uint16_t w3_u16x2 __attribute__((ext_vector_type(2)));

// lanes with `on` add their coded bits at key[j]; v[j] = the Counter before this lane's own update
__device__ __forceinline__ void atomic_round(uint32_t *tbl, uint32_t c0, const uint32_t key[8], bool on, uint32_t v[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t inc = on ? (((c0 >> (7 - j)) & 1u) ? 0x10000u : 1u) : 0u;
        v[j] = __hip_atomic_fetch_add(&tbl[j * 256 + key[j]], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// any returned count at or above the limit (wave-uniform)
__device__ __forceinline__ bool atomic_round_hot(const uint32_t v[8]) {
    w3_u16x2 m = __builtin_bit_cast(w3_u16x2, v[0]);
#pragma unroll
    for (int j = 1; j < 8; j++) m = __builtin_elementwise_max(m, __builtin_bit_cast(w3_u16x2, v[j]));
    return __ballot(max((uint32_t)m.x, (uint32_t)m.y) >= W3_ATOMIC_LIM) != 0ull;
}

// Per-device check of the lane-order property atomic_round relies on: 64 wavefronts x 32 rounds of returning adds on keys
// from one shared hash (all lanes one address ... 2048 addresses); the host replays them in lane order (twophase_lds_order_ok).
__host__ __device__ __forceinline__ uint32_t lds_order_hash(uint32_t wave, uint32_t round, uint32_t lane) {
    uint32_t x = (wave * 64u + round) * 64u + lane + 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t lds_order_key(uint32_t wave, uint32_t h) {
    const uint32_t kind = wave & 3u;
    return kind == 0 ? 77u : kind == 1 ? (h & 7u) * 32u + 1u : kind == 2 ? (h & 255u) : (h & 2047u);
}
__global__ void __launch_bounds__(64) k_lds_order_selftest(uint32_t *old) {
    __shared__ uint32_t tbl[2048];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 2048u; i += 64u) tbl[i] = 0u;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t r = 0; r < 32u; r++) {
        const uint32_t h = lds_order_hash(blockIdx.x, r, lane);
        old[(blockIdx.x * 32u + r) * 64u + lane] =
            __hip_atomic_fetch_add(&tbl[lds_order_key(blockIdx.x, h)], (h >> 20) & 1u ? 0x10000u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__device__ __forceinline__ uint4 pack_p(const uint32_t p[8]) {
    return make_uint4(p[0] | (p[1] << 16), p[2] | (p[3] << 16), p[4] | (p[5] << 16), p[6] | (p[7] << 16));
}

// OpinionMixer2 merge of a later leaf into the running stream: replace only on
// strictly larger distance (ties keep the earlier = left leaf).
__device__ __forceinline__ uint32_t mix_pair(uint32_t cur, uint32_t nw) {
    const uint32_t c_lo = cur & 0xFFFFu, c_hi = cur >> 16, n_lo = nw & 0xFFFFu, n_hi = nw >> 16;
    const uint32_t lo = opinion_dist(n_lo) > opinion_dist(c_lo) ? n_lo : c_lo;
    const uint32_t hi = opinion_dist(n_hi) > opinion_dist(c_hi) ? n_hi : c_hi;
    return lo | (hi << 16);
}
__device__ __forceinline__ uint4 mix_p(uint4 cur, uint4 nw) {
    return make_uint4(mix_pair(cur.x, nw.x), mix_pair(cur.y, nw.y), mix_pair(cur.z, nw.z), mix_pair(cur.w, nw.w));
}

#include "w3_window.h"   // window_head / load_window / wave_window: the input-window loads (plain C++, also compiled for the host by tests/test_window_loads.py)

// ---------------------------------------------------------------------------
// H <= 8, time order.  KEYS: key bytes come from args.keys (ACHistory leaves).
// ---------------------------------------------------------------------------
template <int H, bool KEYS, int NW = 1>
__global__ void __launch_bounds__(64 * NW) k_predict_small(PredictArgs a) {
    __shared__ uint32_t tbl_[NW][8 * 256];
    __shared__ uint32_t st_w_[NW][W3_PF * 64];                 // operand staging (one batch of rounds)
    __shared__ uint2 st_k_[NW][KEYS ? W3_PF * 64 : 1];
    W3_HALF_CU_PAD(NW, sizeof(tbl_) + sizeof(st_w_) + sizeof(st_k_))
    const int lane = threadIdx.x & 63;
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t *tbl = tbl_[wv], *st_w = st_w_[wv];
    uint2 *st_k = st_k_[wv];
    constexpr uint32_t KM = (1u << H) - 1u;
    for (uint32_t b = blockIdx.x * NW + wv; b < a.nblocks; b += gridDim.x * NW) {
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
        const uint8_t *blk = a.in + off;
#pragma unroll
        for (int k = 0; k < 32; k++) tbl[k * 64 + lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        bool exact = (a.dbg_flags & 2u) != 0u;   // ballot rounds (LDS-atomic order self-test failed, or a count nears 65535)
        // Operands travel in batches of W3_PF rounds: the loads of batch k+1 are issued at the start of batch k (unconditional,
        // index clamped) and parked in LDS at its end, and the rounds read them from LDS.  That keeps loads and stores from being
        // in flight together inside a batch: with both pending hipcc waits vmcnt(0)/(1) EVERY round, i.e. for the round's own
        // 16-byte store to complete (gfx9 counts loads and stores in one counter and LLVM assumes they retire out of order).
        const uint32_t first = window_head(off, 3u);   // (the name is history: 0 = the input's first block)
        const uint32_t last = len - 1u;
        uint32_t nw[W3_PF]; uint2 nk[W3_PF];
#pragma unroll
        for (int r = 0; r < W3_PF; r++) {
            const uint32_t ic = min((uint32_t)(r * 64 + lane), last);
            if constexpr (KEYS) { nk[r] = a.keys[off + ic]; nw[r] = blk[ic]; } else { nw[r] = load_window(blk, ic, first); nk[r] = make_uint2(0, 0); }
        }
        for (uint32_t bbase = 0; bbase < len; bbase += 64u * W3_PF) {
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < W3_PF; r++) { st_w[r * 64 + lane] = nw[r]; if constexpr (KEYS) st_k[r * 64 + lane] = nk[r]; }
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < W3_PF; r++) {
                const uint32_t ic = min(bbase + (uint32_t)((W3_PF + r) * 64 + lane), last);
                if constexpr (KEYS) { nk[r] = a.keys[off + ic]; nw[r] = blk[ic]; } else nw[r] = load_window(blk, ic, first);
            }
#pragma unroll 1
          for (uint32_t rr = 0; rr < W3_PF; rr++) {
            const uint32_t base = bbase + rr * 64u;
            if (base >= len) break;
            const uint32_t i = base + lane;
            const bool valid = i < len;
            const uint32_t w = st_w[rr * 64u + lane];
            uint2 k8 = make_uint2(0, 0);
            if constexpr (KEYS) k8 = st_k[rr * 64u + lane];
            uint32_t c0 = 0, key[8], p[8];
            if constexpr (KEYS) {
                c0 = valid ? w : 0u;
#pragma unroll
                for (int j = 0; j < 8; j++) key[j] = ((j < 4 ? k8.x : k8.y) >> (8 * (j & 3))) & KM;
            } else {
                const uint32_t wv = valid ? w : 0u;
                c0 = wv & 0xFFu;
#pragma unroll
                for (int j = 0; j < 8; j++) key[j] = ((wv & 0xFFFFu) >> (8 - j)) & KM;
            }
            if (!exact) {
                uint32_t v[8];
                __asm__ volatile("" ::: "memory");
                atomic_round(tbl, c0, key, valid, v);
                __asm__ volatile("" ::: "memory");
                if ((a.dbg_flags & 8u) && base == 64u && (a.fault_block == 0xFFFFFFFFu || a.fault_block == b)) v[0] = (uint32_t)__shfl_xor((int)v[0], 1, 64);   // (test hook: a mis-ordered add)
#pragma unroll
                for (int j = 0; j < 8; j++) p[j] = counter_p_packed(v[j]);
                exact = atomic_round_hot(v);
            } else {
                uint64_t M[8];
                if constexpr (KEYS) match_keys<H>(valid ? k8 : make_uint2(0, 0), M);
                else match_windows<H>((valid ? w : 0u) & 0xFFFFu, M);
                const uint64_t seg = __ballot(valid);
                uint32_t fin[8], wm;
                rank_round<false>(c0, key, M, seg, valid, true, tbl, p, fin, wm);
            }
            {
                uint4 *dst = valid ? a.P + (off + i) : a.sink + lane;   // unconditional store (see PredictArgs::sink)
                *dst = pack_p(p);
            }
          }
        }
    }
}

// ---------------------------------------------------------------------------
// H == 16 / 24: stable partition of the positions by c1 (and c2), then rank.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, uint32_t *total) {
    const int lane = threadIdx.x & 63;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    *total = __shfl(inc, 63, 64);
    return inc - v;
}

// (Fallback when the LDS-add lane order is not available, and for an order-2 leaf with no Order1 leaf ahead of it; the default
// is k_partition8 below.)  Stable LSD partition of the block's positions by the previous byte(s), 4 bits per pass.
// 16 bins keep only 16 open output lines per wave, so the scattered position writes merge
// into full lines in L2 (256 bins x 4096 waves overflowed the L2s: every 4-byte store became
// its own HBM transaction).  hist[pass][16] is filled in ONE time-ordered sweep up front.

template <int NPASS, bool C2ONLY = false>
__device__ __forceinline__ void partition_hist(const uint8_t *blk, uint32_t len, uint32_t first, uint32_t *hist) {
    const int lane = threadIdx.x & 63;
    const uint64_t gt = lane_gt_mask();
    if (lane < 16 * NPASS) hist[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    // Rounds are short, so a round-at-a-time loop would sit out one memory latency (~4-6k cycles under
    // load) per round: the loads of W3_PF rounds are issued together, one batch ahead.
    uint32_t wn[W3_PF];
    const uint32_t last = len - 1u;
#pragma unroll
    for (int r = 0; r < W3_PF; r++) wn[r] = load_window(blk, min(r * 64u + lane, last), first);
    for (uint32_t base = 0; base < len; base += 64u * W3_PF) {
        uint32_t wc[W3_PF];
#pragma unroll
        for (int r = 0; r < W3_PF; r++) wc[r] = wn[r];
#pragma unroll
        for (int r = 0; r < W3_PF; r++) wn[r] = load_window(blk, min(base + (W3_PF + r) * 64u + lane, last), first);
#pragma unroll
        for (int r = 0; r < W3_PF; r++) {
            const uint32_t i = base + r * 64u + lane;
            if (base + r * 64u >= len) break;
            const bool valid = i < len;
            const uint32_t kb = valid ? (wc[r] >> 8) & 0xFFFFu : 0u;   // c1 | c2 << 8 (zeros before the block start)
            const uint64_t vm = __ballot(valid);
#pragma unroll
            for (int ps = 0; ps < NPASS; ps++) {
                // pass order (LSD): c2 low, c2 high, c1 low, c1 high  /  c1 low, c1 high
                //             C2ONLY: c2 low, c2 high (the records arrive already sorted by c1)
                const uint32_t d = C2ONLY ? (kb >> (8 + 4 * ps)) & 15u
                                 : NPASS == 4 ? ((ps < 2 ? (kb >> 8) : kb) >> (4 * (ps & 1))) & 15u : (kb >> (4 * ps)) & 15u;
                uint64_t m = vm;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool mybit = (d >> k) & 1u;
                    const uint64_t B = __ballot(mybit);
                    m &= mybit ? B : ~B;
                }
                if (valid && (m & gt) == 0ull) hist[ps * 16 + d] += (uint32_t)__popcll(m);
                __asm__ volatile("" ::: "memory");
            }
        }
    }
    // exclusive scan inside each pass's 16 bins
    uint32_t v = lane < 16 * NPASS ? hist[lane] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int dd = 1; dd < 16; dd <<= 1) {
        uint32_t o = __shfl_up(inc, dd, 64);
        if ((lane & 15) >= dd) inc += o;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 16 * NPASS) hist[lane] = inc - v;
    __builtin_amdgcn_wave_barrier();
}

// One stable 16-way pass.  Elements travel as RECORDS (position, window bytes c0..c3), so
// no pass ever gathers from the input block again: with 4096 waves in flight the blocks do
// not stay in L2 and every gathered byte cost a 64-byte fetch (212 GB of FETCH per GB input).
template <bool FROM_INPUT>
__device__ __forceinline__ void partition_pass4(const uint8_t *blk, uint32_t len, uint32_t first, uint32_t back, uint32_t shift, const uint2 *src, uint2 *dst, uint32_t *bins) {
    const int lane = threadIdx.x & 63;
    const uint64_t gt = lane_gt_mask();
    const uint32_t last = len - 1u;
    uint2 rn[W3_PF];
#pragma unroll
    for (int r = 0; r < W3_PF; r++) {
        const uint32_t e = min(r * 64u + lane, last);
        if constexpr (FROM_INPUT) rn[r] = make_uint2(e, load_window(blk, e, first)); else rn[r] = src[e];
    }
    for (uint32_t base = 0; base < len; base += 64u * W3_PF) {
        uint2 rc[W3_PF];
#pragma unroll
        for (int r = 0; r < W3_PF; r++) rc[r] = rn[r];
#pragma unroll
        for (int r = 0; r < W3_PF; r++) {
            const uint32_t e = min(base + (W3_PF + r) * 64u + lane, last);
            if constexpr (FROM_INPUT) rn[r] = make_uint2(e, load_window(blk, e, first)); else rn[r] = src[e];
        }
#pragma unroll
        for (int r = 0; r < W3_PF; r++) {
            const uint32_t e = base + r * 64u + lane;
            if (base + r * 64u >= len) break;
            const bool valid = e < len;
            const uint2 rec = rc[r];
            const uint32_t d = valid ? (rec.y >> (8u * back + shift)) & 15u : 0u;
            uint64_t m = __ballot(valid);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool mybit = (d >> k) & 1u;
                const uint64_t B = __ballot(mybit);
                m &= mybit ? B : ~B;
            }
            if (valid) {
                const uint32_t bs = bins[d];
                dst[bs + mbcnt64(m)] = rec;
                if ((m & gt) == 0ull) bins[d] = bs + (uint32_t)__popcll(m);
            }
            __asm__ volatile("" ::: "memory");
        }
    }
}

#define W3_SLICES 64u   // rank jobs per block (sorted range cut at group boundaries)

// k_partition<NBYTES>: one wavefront per block sorts the block's records stably by the group key
// (NBYTES 1: c1 — Order1; 2: (c1,c2) — OrderN(27,3)) into a.rec, and cuts the sorted range into
// W3_SLICES slices at group boundaries (a.splits) for k_rank_sorted.
// NBYTES 3: the (c1,c2) grouping built from an Order1 leaf's records (a.rec_src, already sorted by c1): two more
// passes over c2 instead of four from scratch.  The groups come out c2-major; k_rank_sorted only needs them contiguous
// and time-ordered inside, which any stable pass order gives.
template <int NBYTES>
__global__ void __launch_bounds__(64) k_partition(PredictArgs a) {
    __shared__ uint32_t hist[64];
    const int lane = threadIdx.x;
    uint2 *perm_a = reinterpret_cast<uint2 *>(a.perm) + (uint64_t)blockIdx.x * 2u * a.block_size;
    uint2 *perm_b = perm_a + a.block_size;
    for (uint32_t b = blockIdx.x; b < a.nblocks; b += gridDim.x) {
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
        const uint8_t *blk = a.in + off;
        uint2 *out = a.rec + off;
        unsigned long long t_prev = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
        const uint32_t first = window_head(off, 3u);   // (the name is history: 0 = the input's first block)
        if constexpr (NBYTES == 3) partition_hist<2, true>(blk, len, first, hist); else partition_hist<2 * NBYTES>(blk, len, first, hist);
        W3_STAMP(0);
        if constexpr (NBYTES == 3) {
            partition_pass4<false>(blk, len, first, 2, 0, a.rec_src + off, perm_a, hist);
            __threadfence_block();
            W3_STAMP(1);
            partition_pass4<false>(blk, len, first, 2, 4, perm_a, out, hist + 16);
        } else if constexpr (NBYTES == 1) {
            partition_pass4<true>(blk, len, first, 1, 0, nullptr, perm_a, hist);
            __threadfence_block();
            W3_STAMP(1);
            partition_pass4<false>(blk, len, first, 1, 4, perm_a, out, hist + 16);
        } else {
            partition_pass4<true>(blk, len, first, 2, 0, nullptr, perm_a, hist);            // LSD: minor key c2 first
            __threadfence_block();
            W3_STAMP(1);
            partition_pass4<false>(blk, len, first, 2, 4, perm_a, perm_b, hist + 16);
            __threadfence_block();
            partition_pass4<false>(blk, len, first, 1, 0, perm_b, perm_a, hist + 32);
            __threadfence_block();
            partition_pass4<false>(blk, len, first, 1, 4, perm_a, out, hist + 48);
        }
        __threadfence_block();
        W3_STAMP(2);
        // slice boundaries: the first group start at or after s*len/W3_SLICES
        uint32_t *sp = a.splits + (uint64_t)b * (W3_SLICES + 1u);
        uint32_t prev = 0u;
        if (lane == 0) { sp[0] = 0u; sp[W3_SLICES] = len; }
        for (uint32_t sl = 1; sl < W3_SLICES; sl++) {
            uint32_t start = max((uint32_t)((uint64_t)sl * len / W3_SLICES), prev), found = len;
            for (uint32_t base = start; base < len; base += 64) {
                const uint32_t e = base + lane;
                bool head = false;
                if (e < len) {
                    const uint32_t g = NBYTES == 1 ? ((out[e].y >> 8) & 0xFFu) : ((out[e].y >> 8) & 0xFFFFu);
                    const uint32_t gp = e ? (NBYTES == 1 ? ((out[e - 1].y >> 8) & 0xFFu) : ((out[e - 1].y >> 8) & 0xFFFFu)) : 0xFFFFFFFFu;
                    head = g != gp;
                }
                const uint64_t hm = __ballot(head);
                if (hm) { found = base + (uint32_t)(__ffsll((long long)hm) - 1); break; }
            }
            if (lane == 0) sp[sl] = found;
            prev = found;
        }
        W3_STAMP(4);
    }
}

// ---------------------------------------------------------------------------
// k_partition8<MODE>: the same stable partition in ONE 8-bit pass per key byte.  256 open output runs per wave would turn
// every 8-byte record store into its own memory transaction (the reason for the 4-bit passes above), so records go through
// LDS tiles: a tile of 2048 records is counting-sorted by the digit inside LDS — the slot of a record is ONE returning LDS add
// on its bin's cursor, stable because the adds are lane-ordered (atomic_round's property, same self-test) and the rounds are
// issued in time order — and then copied out bin run by bin run, consecutive lanes to consecutive addresses.
//   MODE 1: records from the input, key c1 (Order1).   MODE 3: records from a.rec_src (sorted by c1), key c2 (order 2 refined).
// ---------------------------------------------------------------------------
#define W3_P8_TILE 2048u
#define W3_P8_ROUNDS (W3_P8_TILE / 64u)

__device__ __forceinline__ void wave_excl_scan_256(uint32_t *cnt, uint32_t *excl, uint32_t *excl2) {
    // 256 counts, 4 consecutive bins per lane
    const int lane = threadIdx.x & 63;
    const uint32_t c0 = cnt[4 * lane], c1 = cnt[4 * lane + 1], c2 = cnt[4 * lane + 2], c3 = cnt[4 * lane + 3];
    uint32_t tot;
    const uint32_t base = wave_excl_scan_u32(c0 + c1 + c2 + c3, &tot);
    excl[4 * lane] = base; excl[4 * lane + 1] = base + c0; excl[4 * lane + 2] = base + c0 + c1; excl[4 * lane + 3] = base + c0 + c1 + c2;
    if (excl2) { excl2[4 * lane] = base; excl2[4 * lane + 1] = base + c0; excl2[4 * lane + 2] = base + c0 + c1; excl2[4 * lane + 3] = base + c0 + c1 + c2; }
}

template <int MODE, int NW = 1>
__global__ void __launch_bounds__(64 * NW) k_partition8(PredictArgs a) {
    __shared__ uint2 tile_[NW][W3_P8_TILE];
    __shared__ uint32_t cnt_[NW][4][256];   // gcur, tcnt, tstart, tcur
    W3_HALF_CU_PAD(NW, sizeof(tile_) + sizeof(cnt_))
    const int lane = threadIdx.x & 63;
    const uint32_t wv = threadIdx.x >> 6;
    uint2 *tile = tile_[wv];
    uint32_t *gcur = cnt_[wv][0], *tcnt = cnt_[wv][1], *tstart = cnt_[wv][2], *tcur = cnt_[wv][3];
    constexpr uint32_t KSH = MODE == 1 ? 8u : 16u;   // digit = window byte c1 / c2
    for (uint32_t b = blockIdx.x * NW + wv; b < a.nblocks; b += gridDim.x * NW) {
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
        const uint8_t *blk = a.in + off;
        const uint2 *src = MODE == 3 ? a.rec_src + off : nullptr;
        uint2 *out = a.rec + off;
        const uint32_t first = window_head(off, 3u);   // (the name is history: 0 = the input's first block)
        const uint32_t last = len - 1u;
        unsigned long long t_prev = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;   // W3_OPT_DEBUG_STAMPS: slots 0 histogram, 1 tile load + count + scan, 2 scatter into the tile, 5 copy out, 4 splits
        // digit counts of the whole block.  digit(pos) = byte[pos - K] (K = 1: c1, 2: c2; zeros before the block start), so
        // this is the byte histogram of bytes [0, len - K) plus K zeros: 16 input bytes per lane and load
        constexpr uint32_t K = MODE == 1 ? 1u : 2u;
#pragma unroll
        for (int k = 0; k < 4; k++) tcnt[k * 64 + lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        __asm__ volatile("" ::: "memory");
        {
            const uint32_t nbytes = len > K ? len - K : 0u;
            if (lane == 0) tcnt[0] = min(K, len);
            __builtin_amdgcn_wave_barrier();
            __asm__ volatile("" ::: "memory");
            if (len >= 16u) {
                const uint32_t cs_max = len - 16u;   // chunks are clamped into the block; bytes before the chunk's own start are skipped
                constexpr int HP = 4;
                uint4 qn[HP];
#pragma unroll
                for (int r = 0; r < HP; r++) __builtin_memcpy(&qn[r], blk + min((uint32_t)(r * 1024 + lane * 16), cs_max), 16);
                for (uint32_t base = 0; base < nbytes; base += 1024u * HP) {
                    uint4 qc[HP];
#pragma unroll
                    for (int r = 0; r < HP; r++) qc[r] = qn[r];
#pragma unroll
                    for (int r = 0; r < HP; r++) __builtin_memcpy(&qn[r], blk + min(base + (uint32_t)((HP + r) * 1024 + lane * 16), cs_max), 16);
#pragma unroll
                    for (int r = 0; r < HP; r++) {
                        const uint32_t want = base + (uint32_t)(r * 1024 + lane * 16), cs = min(want, cs_max);
                        const uint32_t w4[4] = {qc[r].x, qc[r].y, qc[r].z, qc[r].w};
#pragma unroll
                        for (int q = 0; q < 16; q++) {
                            const uint32_t pos = cs + (uint32_t)q;
                            if (pos >= want && pos < nbytes)
                                __hip_atomic_fetch_add(&tcnt[(w4[q >> 2] >> (8 * (q & 3))) & 0xFFu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            } else {
                for (uint32_t pos = lane; pos < nbytes; pos += 64u)
                    __hip_atomic_fetch_add(&tcnt[blk[pos]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        wave_excl_scan_256(tcnt, gcur, nullptr);
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        W3_STAMP(0);
        for (uint32_t t0 = 0; t0 < len; t0 += W3_P8_TILE) {
            const uint32_t tlen = min(W3_P8_TILE, len - t0);
            uint2 rec[W3_P8_ROUNDS];   // (loading a tile ahead, while the one before is sorted and copied out, changes nothing: measured)
#pragma unroll
            for (uint32_t r = 0; r < W3_P8_ROUNDS; r++) {
                const uint32_t e = min(t0 + r * 64u + lane, last);
                if constexpr (MODE == 1) rec[r] = make_uint2(e, load_window(blk, e, first)); else rec[r] = src[e];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) tcnt[k * 64 + lane] = 0u;
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t r = 0; r < W3_P8_ROUNDS; r++)
                if (r * 64u + lane < tlen) __hip_atomic_fetch_add(&tcnt[(rec[r].y >> KSH) & 0xFFu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            wave_excl_scan_256(tcnt, tstart, tcur);
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            W3_STAMP(1);
            // stable scatter into the tile: rounds in time order, lanes in order inside the returning add
#pragma unroll
            for (uint32_t r = 0; r < W3_P8_ROUNDS; r++) {
                if (r * 64u + lane < tlen) {
                    const uint32_t slot = __hip_atomic_fetch_add(&tcur[(rec[r].y >> KSH) & 0xFFu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    tile[slot] = rec[r];
                }
            }
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            W3_STAMP(2);
            // copy out: element k of the sorted tile belongs to bin d at run offset k - tstart[d]
#pragma unroll
            for (uint32_t r = 0; r < W3_P8_ROUNDS; r++) {
                const uint32_t k = r * 64u + lane;
                if (k < tlen) {
                    const uint2 rc = tile[k];
                    const uint32_t d = (rc.y >> KSH) & 0xFFu;
                    out[gcur[d] + (k - tstart[d])] = rc;
                }
            }
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++) gcur[k * 64 + lane] += tcnt[k * 64 + lane];
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            W3_STAMP(5);
        }
        __threadfence_block();
        // slice boundaries: the first group start at or after s*len/W3_SLICES (lane sl finds boundary sl; W3_SLICES == 64)
        uint32_t *sp = a.splits + (uint64_t)b * (W3_SLICES + 1u);
        static_assert(W3_SLICES == 64u, "one lane per slice boundary");
        const uint32_t ideal = (uint32_t)((uint64_t)lane * len / W3_SLICES);
        uint32_t found;
        if constexpr (MODE == 1) {
            // groups = the digit's bins, and after the last tile gcur[d] is the END of bin d = the start of the next group: no
            // record has to be read back.  (The 63 serial searches through the block's own output for the first group head — two
            // dependent global loads each — were more than half of this kernel's time: W3_OPT_DEBUG_STAMPS.)
            uint32_t lo = 0u, hi = 256u;   // smallest d in [0, 256] with start(d) >= ideal; start(0) = 0, start(d) = gcur[d - 1]
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t st = mid == 0u ? 0u : gcur[mid - 1u];
                if (st >= ideal) hi = mid; else lo = mid + 1u;
            }
            found = lo == 0u ? 0u : gcur[lo - 1u];
        } else {
            // groups = (c2, c1) pairs inside the c2 bins (bin-aligned slices are too coarse here: k_rank_sorted<2> 13 -> 17 ms,
            // measured).  The searches run EIGHT AT A TIME: the 64 records from each of eight ideal points are loaded together,
            // then looked through; a group longer than that is walked on 64 records at a time (once: the boundaries behind it
            // inside the same group take the same answer).
            found = 0xFFFFFFFFu;
            for (uint32_t s0 = 1; s0 < W3_SLICES; s0 += 8u) {
                uint32_t gy[8], gpy[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t st = readlane_u32(ideal, (int)min(s0 + (uint32_t)k, W3_SLICES - 1u));
                    const uint32_t e = min(st + lane, last);
                    gy[k] = out[e].y; gpy[k] = out[e ? e - 1u : 0u].y;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t sl = s0 + (uint32_t)k;
                    if (sl >= W3_SLICES) break;
                    const uint32_t st = readlane_u32(ideal, (int)sl);
                    const uint32_t e = st + lane;
                    const bool head = e < len && (e == 0u || ((gy[k] >> 8) & 0xFFFFu) != ((gpy[k] >> 8) & 0xFFFFu));
                    const uint64_t hm = __ballot(head);
                    uint32_t f;
                    if (hm) f = st + (uint32_t)(__ffsll((long long)hm) - 1);
                    else {
                        const uint32_t before = readlane_u32(found, (int)sl - 1);
                        f = len;
                        if (sl > 1u && before != 0xFFFFFFFFu && before >= st) f = before;   // still inside the group the boundary before skipped
                        else for (uint32_t base = st + 64u; base < len; base += 64u) {
                            const uint32_t e2 = base + lane;
                            const bool h2 = e2 < len && ((out[e2].y >> 8) & 0xFFFFu) != ((out[e2 - 1u].y >> 8) & 0xFFFFu);
                            const uint64_t hm2 = __ballot(h2);
                            if (hm2) { f = base + (uint32_t)(__ffsll((long long)hm2) - 1); break; }
                        }
                    }
                    if (lane == sl) found = f;
                }
            }
        }
        if (lane == 0) found = 0u;
        // boundaries never decrease (a later ideal point lies in the same or a later group)
        sp[lane] = found;
        if (lane == 0) sp[W3_SLICES] = len;
        W3_STAMP(4);
        if (a.dbg && lane == 0) atomicAdd(&a.dbg[6], 1ull);
    }
}

// k_rank_sorted<NBYTES>: job = (block, slice).  A persistent grid of 2048 wavefronts walks the jobs in
// block-major order, so only ~2048/W3_SLICES = 32 blocks (more while a block's largest group is still running) are being
// scattered into at any time: their
// P regions (1 MiB each) then stay in the 256 MiB Infinity Cache, where the eight partial 16-byte
// writes every 128-byte line receives merge (3.2x cheaper than with 4096 blocks live; see
// profiles/r1_ubench_partial_line_merge_vs_footprint.txt).
// PF: rounds per operand batch (8: 12 KiB of LDS per wavefront; 4: 10 KiB, so that EIGHT wavefronts fit the half of a CU)
template <int NBYTES, int NW = 1, int PF = W3_PF>
__global__ void __launch_bounds__(64 * NW) k_rank_sorted(PredictArgs a) {
    __shared__ uint32_t tbl_[NW][8 * 256];
    __shared__ uint2 st_r_[NW][PF * 64];   // record staging (one batch of rounds)
    W3_HALF_CU_PAD(NW, sizeof(tbl_) + sizeof(st_r_))
    const int lane = threadIdx.x & 63;
    uint32_t *tbl = tbl_[threadIdx.x >> 6];
    uint2 *st_r = st_r_[threadIdx.x >> 6];
    const uint32_t njobs = a.nblocks * W3_SLICES;
    // jobs are handed out in block-major order from one counter: slices are very uneven (a block's biggest
    // group is one slice), and in-order hand-out keeps the set of blocks being scattered into small.
    // (Round 2 tried one queue per XCD — block b to queue b % 8, waves pulling from the queue of HW_REG_XCC_ID — so that the
    // eight 16-byte pieces of a line would meet in ONE L2: WRITE_SIZE went UP, 31.4 -> 34.6 GB and 32.4 -> 44.5 GB per launch,
    // and the phase from 47.5 to 48.5 ms: four blocks' streams fill an XCD's 4 MiB L2, lines are evicted before their
    // pieces meet, and the partial writes no longer merge in the Infinity Cache behind a single stream of jobs either.)
    for (;;) {
        uint32_t job = 0;
        if (lane == 0) job = atomicAdd(a.job_counter, 1u);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) break;
        const uint32_t b = job / W3_SLICES, sl = job % W3_SLICES;
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint32_t *sp = a.splits + (uint64_t)b * (W3_SLICES + 1u);
        const uint32_t lo = sp[sl], hi = sp[sl + 1];
        if (lo >= hi) continue;
        const uint32_t len = hi - lo;
        const uint2 *perm = a.rec + off + lo;
        unsigned long long t_prev = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
        // The table describes ONE group at a time: the group that is still open at the end of a round.
        // Groups that start and end inside a round never touch it (their Counters start new).
#pragma unroll
        for (int k = 0; k < 32; k++) tbl[k * 64 + lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        bool exact = (a.dbg_flags & 2u) != 0u;   // ballot rounds only (see atomic_round)
        bool dirty = false;              // table holds states of group open_g
        uint32_t open_g = 0xFFFFFFFFu;   // group the table describes (also: group of the previous round's last element)
        // records travel in batches of PF rounds through LDS (see k_predict_small: no load is in flight beside the stores)
        const uint32_t last = len - 1u;
        uint2 rn[PF];
#pragma unroll
        for (int r = 0; r < PF; r++) rn[r] = perm[min((uint32_t)(r * 64 + lane), last)];
        for (uint32_t bbase = 0; bbase < len; bbase += 64u * (uint32_t)PF) {
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < PF; r++) st_r[r * 64 + lane] = rn[r];
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < PF; r++) rn[r] = perm[min(bbase + (uint32_t)((PF + r) * 64 + lane), last)];
#pragma unroll 1
          for (uint32_t rr = 0; rr < (uint32_t)PF; rr++) {
            const uint32_t base = bbase + rr * 64u;
            if (base >= len) break;
            const uint32_t e = base + lane;
            const bool valid = e < len;
            const uint2 r_c = st_r[rr * 64u + lane];
            const uint32_t i = r_c.x; const uint32_t w = valid ? r_c.y : 0u;
            const uint32_t c0 = w & 0xFFu;
            // group id and the byte that supplies the low key bits
            const uint32_t g = NBYTES == 1 ? ((w >> 8) & 0xFFu) : ((w >> 8) & 0xFFFFu);
            const uint32_t ck = NBYTES == 1 ? ((w >> 16) & 0xFFu) : (w >> 24);
            const uint32_t w16 = (ck << 8) | c0;
            // segment = lanes of my group: groups are contiguous in the sorted order
            uint32_t gprev = __shfl_up(g, 1, 64);
            if (lane == 0) gprev = open_g;
            const uint64_t vm = __ballot(valid);
            const uint64_t heads = __ballot(valid && g != gprev) | 1ull;  // lane 0 opens a segment (same tag if the group continues)
            const uint64_t hle = heads & (lane_lt_mask() | (1ull << lane));
            const int start = 63 - __clzll((long long)hle);
            const uint64_t hgt = heads & lane_gt_mask();
            const uint64_t below_end = hgt ? ((1ull << (__ffsll((long long)hgt) - 1)) - 1ull) : ~0ull;
            const uint64_t seg = below_end & ~((1ull << start) - 1ull) & vm;
            uint32_t key[8], p[8];
#pragma unroll
            for (int j = 0; j < 8; j++) key[j] = (w16 >> (8 - j)) & 0xFFu;
            uint32_t *vt = tbl;
            // the group of the round's last valid element stays open into the next round: its states stay in the table
            const int lastlane = 63 - __clzll((long long)vm);
            const uint32_t g_last = readlane_u32(g, lastlane);
            uint64_t hm = heads & vm;   // segment starts
            if (!exact && (uint32_t)__popcll(hm) <= a.maxseg) {
                // LDS-atomic rounds (atomic_round), one segment (= group) after the other; a group that ends inside the
                // round leaves the table empty again: its own lanes zero what they touched, or, when it came in from
                // earlier rounds, the whole table is cleared
                uint32_t v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = 0u;
                __asm__ volatile("" ::: "memory");
                if (dirty && readlane_u32(g, 0) != open_g) {
#pragma unroll
                    for (int k = 0; k < 32; k++) vt[k * 64 + lane] = 0u;
                    dirty = false;
                }
                bool carried = dirty;
                while (hm) {
                    const int st = __ffsll((long long)hm) - 1;
                    hm &= hm - 1ull;
                    const int en = hm ? __ffsll((long long)hm) - 1 : 64;
                    const bool in = valid && lane >= st && lane < en;
                    __asm__ volatile("" ::: "memory");
                    if (in) atomic_round(vt, c0, key, true, v);
                    __asm__ volatile("" ::: "memory");
                    if (hm) {
                        if (carried) {
#pragma unroll
                            for (int k = 0; k < 32; k++) vt[k * 64 + lane] = 0u;
                        } else if (in) {
#pragma unroll
                            for (int j = 0; j < 8; j++) vt[j * 256 + key[j]] = 0u;
                        }
                    }
                    carried = false;
                }
                __asm__ volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; j++) p[j] = counter_p_packed(v[j]);
                exact = atomic_round_hot(v);
            } else {
                uint32_t fin[8], wm;
                uint64_t M[8];
                match_windows<8>(w16, M);
                rank_round<true>(c0, key, M, seg, valid, g == open_g, vt, p, fin, wm);
                if (g_last != open_g && dirty) {
#pragma unroll
                    for (int k = 0; k < 32; k++) vt[k * 64 + lane] = 0u;
                    dirty = false;
                }
                __builtin_amdgcn_wave_barrier();
                if (valid && g == g_last) {
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if ((wm >> j) & 1u) vt[j * 256 + key[j]] = fin[j];
                }
                __asm__ volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            dirty = true;
            open_g = g_last;
            {   // 16-byte scatter, write-only; unconditional (see PredictArgs::sink)
                uint4 *dst = (valid && !(a.dbg_flags & 1u)) ? a.P + (off + ((a.dbg_flags & 4u) ? (i & 0xFFFu) : i)) : a.sink + lane;
                *dst = pack_p(p);
            }
          }
        }
        W3_STAMP(3);
        if (a.dbg && lane == 0 && sl == 0) atomicAdd(&a.dbg[7], 1ull);
    }
}

// ---------------------------------------------------------------------------
// HuffHistory keys (history/huff_history.rs:58-76): one thread per byte position computes the 8 context hashes of its
// bit positions.  compressed_bits before byte i is the concatenation of the codes of the bytes before it (newest lowest),
// cut to 32 bits: walk back until 32 bits are covered (a handful of bytes: codes average 4-6 bits).
// The walk is BOUNDED (W3_HUFF_WALK bytes): a symbol absent from the training buffer has code length 0 (legal through
// HuffHistory::from_tables / w3_huff_table) and does not advance the covered bits, so a long run of such bytes would make
// every thread walk back to the run's start — O(block_size) per thread.  A position whose walk ends uncovered flags its
// block, and k_huffkeys_fix recomputes that block's keys with the reference's own O(1)-per-byte recurrence
// (compressed_bits = compressed_bits << len | code, :60-66), one lane per flagged block.
// ---------------------------------------------------------------------------
#define W3_HUFF_WALK 40u   // 32 bits are covered after at most 32 bytes of non-empty codes
struct HuffKeyArgs { const uint8_t *in; uint64_t n; uint32_t block_size; uint32_t hmask; const w3_huff_table *tb; uint2 *keys;
                     uint32_t *redo;   /* [nblocks] zeroed before k_huffkeys: != 0 -> k_huffkeys_fix recomputes the block */
                     uint32_t nblocks;
                     uint32_t *keys32; /* WIDE form: [8 n] the whole 32-bit hash of every step (w3_predict_wave.h) instead of `keys` */ };

// the 8 hashes of byte c0 at position i of its block, given compressed_bits before it
__device__ __forceinline__ void huff_hashes_of(uint32_t cb, uint32_t c0, const uint16_t *rcode, const uint8_t *rlen, uint32_t (&h)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t rem = (1u << j) | (c0 >> (8 - j));                 // partial byte with a leading 1 (:71-73)
        h[j] = (cb << rlen[rem]) | rcode[rem];                            // :74-75
    }
}
template <bool WIDE>
__device__ __forceinline__ void huff_keys_store(const HuffKeyArgs &a, uint64_t g, uint32_t i, const uint32_t (&h)[8]) {
    if constexpr (WIDE) {
        uint4 *dst = reinterpret_cast<uint4 *>(a.keys32 + g * 8u);
        dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
        dst[1] = make_uint4(h[4], h[5], h[6], h[7]);
    } else {
        uint32_t out[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (i == 0u && j == 0) continue;                              // ctx starts at 0 (ordern_entropy.rs:19)
            out[j >> 2] |= (h[j] & a.hmask & 0xFFu) << (8 * (j & 3));
        }
        a.keys[g] = make_uint2(out[0], out[1]);
    }
}

template <bool WIDE>
__global__ void __launch_bounds__(256) k_huffkeys(HuffKeyArgs a) {
    __shared__ uint16_t s_code[256], s_rcode[256];
    __shared__ uint8_t s_len[256], s_rlen[256];
    s_code[threadIdx.x] = a.tb->code[threadIdx.x]; s_len[threadIdx.x] = a.tb->len[threadIdx.x];
    s_rcode[threadIdx.x] = a.tb->rem_code[threadIdx.x]; s_rlen[threadIdx.x] = a.tb->rem_len[threadIdx.x];
    __syncthreads();
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n) return;
    const uint64_t b = g / a.block_size;
    const uint32_t i = (uint32_t)(g - b * a.block_size);
    const uint8_t *blk = a.in + b * a.block_size;
    uint32_t cb = 0u, have = 0u, k = 1;
    for (; k <= i && have < 32u && k <= W3_HUFF_WALK; k++) {
        const uint32_t byte = blk[i - k];
        cb |= (uint32_t)s_code[byte] << have;
        have += s_len[byte];
    }
    if (have < 32u && k <= i) a.redo[b] = 1u;   // uncovered with bytes left: a run of zero-length codes (benign race: every writer stores 1)
    uint32_t h[8];
    huff_hashes_of(cb, blk[i], s_rcode, s_rlen, h);
    huff_keys_store<WIDE>(a, g, i, h);
}

// one lane per block; only flagged blocks do anything (65,536 serial steps of a few instructions each)
template <bool WIDE>
__global__ void __launch_bounds__(64) k_huffkeys_fix(HuffKeyArgs a) {
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= a.nblocks || a.redo[b] == 0u) return;
    const uint64_t off = (uint64_t)b * a.block_size;
    const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
    uint32_t cb = 0u;
    for (uint32_t i = 0; i < len; i++) {
        const uint32_t c0 = a.in[off + i];
        uint32_t h[8];
        huff_hashes_of(cb, c0, a.tb->rem_code, a.tb->rem_len, h);
        huff_keys_store<WIDE>(a, off + i, i, h);
        const uint32_t l = a.tb->len[c0];
        cb = (l < 32u ? cb << l : 0u) | a.tb->code[c0];
    }
}

// ---------------------------------------------------------------------------
// Sampled verification of the LDS-add rounds (w3_twophase.h, twophase_verify): S blocks of the input are gathered into a
// compact buffer, predicted a second time with the ballot rounds (exact by construction), and the streams compared.
// Sampled block s = block s * nb_full / S + rot (full-length blocks only; rot < nb_full / S rotates the sample from call to call,
// so that after nb_full / S calls every block has been in it once).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_blocks(const uint8_t *in, uint32_t bs, uint32_t nb_full, uint32_t S, uint32_t rot, uint8_t *out) {
    const uint32_t s = blockIdx.y;
    const uint64_t src = (uint64_t)((uint64_t)s * nb_full / S + rot) * bs;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < bs; i += gridDim.x * 256u) out[(uint64_t)s * bs + i] = in[src + i];
}
__global__ void __launch_bounds__(256) k_compare_blocks(const uint4 *main_p, const uint4 *ver_p, uint32_t bs, uint32_t nb_full, uint32_t S, uint32_t rot, uint32_t *mismatch) {
    const uint32_t s = blockIdx.y;
    const uint64_t src = (uint64_t)((uint64_t)s * nb_full / S + rot) * bs;
    uint32_t bad = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < bs; i += gridDim.x * 256u) {
        const uint4 x = main_p[src + i], y = ver_p[(uint64_t)s * bs + i];
        bad |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w);
    }
    if (__ballot(bad != 0u) && (threadIdx.x & 63u) == 0u) atomicAdd(mismatch, 1u);
}

// ---------------------------------------------------------------------------
// k_mix: merge the leaves' streams (in leaf order) into P.  Elementwise, coalesced.
// ---------------------------------------------------------------------------
struct MixArgs {
    const uint4 *src[16];
    int n_src;
    uint4 *P;
    uint64_t n;
};

__global__ void __launch_bounds__(256) k_mix(MixArgs a) {
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < a.n; g += (uint64_t)gridDim.x * blockDim.x) {
        uint4 acc = a.src[0][g];
        for (int l = 1; l < a.n_src; l++) acc = mix_p(acc, a.src[l][g]);
        a.P[g] = acc;
    }
}

// ---------------------------------------------------------------------------
// ACHistory keys: one thread per byte position computes the 8 context hashes of
// its bit positions (history/ac_history.rs:28-46), embarrassingly parallel.
// ---------------------------------------------------------------------------
struct HashArgs {
    const uint8_t *in; uint64_t n; uint32_t block_size; uint32_t max_bits; uint32_t hmask; uint16_t table[8]; uint2 *keys;
    uint4 *lut;   // [65536][8] coder state after the 16 most recent history bits, per bit position (k_achash_lut)
    uint16_t *lut_key;   // [65536][8] the finished key (low 8 bits) | 0x8000 when the hash is complete after those 16 bits
};

__device__ __forceinline__ uint32_t sel8(const uint32_t (&v)[8], uint32_t k) {   // k is wave-uniform: scalar selects, no indexing
    return k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : k == 3 ? v[3] : k == 4 ? v[4] : k == 5 ? v[5] : k == 6 ? v[6] : v[7];
}

// The first 16 coded history bits are the previous two bytes (for bit position 0) or their tail plus the partial byte:
// 65536 x 8 possible prefixes.  Their coder states are tabulated once per call (the idea of the reference's
// ACHistoryCached memo, history/ac_history_cached.rs:31-76, as a dense 8 MiB table that stays in L2 / Infinity Cache);
// every hash starts from its entry at step 16 — with max_bits = 8 most are already complete there.
#define W3_ACHASH_LUT_BITS 16   // (a multiple of 8: ac_history_hash_steps advances 8 history bits at a time)
__global__ void __launch_bounds__(256) k_achash_lut(HashArgs a) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;   // e = v * 8 + j
    if (e >= (8u << W3_ACHASH_LUT_BITS)) return;
    const uint32_t v = e >> 3, j = e & 7u;
    uint32_t p32t[8], rot[8];
#pragma unroll
    for (int k = 0; k < 8; k++) p32t[k] = a.table[k] ? ((uint32_t)a.table[k] << 16) : 1u;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t k = (j + 7u - (uint32_t)r) & 7u;
        rot[r] = k == 0 ? p32t[0] : k == 1 ? p32t[1] : k == 2 ? p32t[2] : k == 3 ? p32t[3] : k == 4 ? p32t[4] : k == 5 ? p32t[5] : k == 6 ? p32t[6] : p32t[7];
    }
    const ACHashState s = ac_history_hash_steps((uint64_t)v, a.max_bits, rot, ac_hash_state_init(a.max_bits), 0, W3_ACHASH_LUT_BITS);
    a.lut[e] = make_uint4(s.x1, s.x2, s.hash, s.meta);
    a.lut_key[e] = (uint16_t)((s.meta >> 31) ? (0x8000u | ((ac_hash_finish(s, a.max_bits) & a.hmask) & 0xFFu)) : 0u);
}

__global__ void __launch_bounds__(256) k_achash(HashArgs a) {
    // Phase 1, per byte position and bit position: the 16-bit prefix table.  Most hashes are complete there (max_bits
    // output bits written); the rest are QUEUED per wavefront (their ids, in LDS).  Phase 2 runs the queue 64 hashes at a
    // time, every lane busy, until the slowest is done — instead of eight loops per wave that each run as long as their
    // slowest lane.  A queued hash is rebuilt from its owner lane's registers (history, byte) and its table entry.
    __shared__ uint16_t q_id[4][512];      // lane << 3 | bit position
    __shared__ uint8_t q_key[4][64][8];    // results of the queued hashes, by (lane, bit position)
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const bool live = g < a.n;
    const uint64_t gc = live ? g : a.n - 1u;
    const uint64_t b = gc / a.block_size;
    const uint32_t i = (uint32_t)(gc - b * a.block_size);
    const uint8_t *blk = a.in + b * a.block_size;
    // the last 64 bits before bit 0 of byte i, newest at bit 0
    uint64_t hist0 = 0;
    for (uint32_t k = 1; k <= 8 && k <= i; k++) hist0 |= (uint64_t)blk[i - k] << (8 * (k - 1));
    const uint32_t c0 = blk[i];
    uint32_t p32t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) p32t[k] = a.table[k] ? ((uint32_t)a.table[k] << 16) : 1u;   // lerp operand, arithmetic_coder.rs:111
    uint32_t out[2] = {0, 0};
    uint32_t qn = 0, pend = 0;   // wave-uniform queue length; this lane's pending bit positions
    uint64_t hist = hist0;
#pragma unroll 1
    for (int j = 0; j < 8; j++) {
        const uint32_t t = i * 8u + j;
        bool pending = false;
        if (live && t != 0u) {   // ctx starts at 0 (ordern_entropy.rs:19)
            const uint32_t le = a.lut_key[((uint32_t)hist & ((1u << W3_ACHASH_LUT_BITS) - 1u)) * 8u + (uint32_t)j];   // 2-byte entry: 1 MiB table
            if (le & 0x8000u) out[j >> 2] |= (le & 0xFFu) << (8 * (j & 3));
            else pending = true;
        }
        const uint64_t pm = __ballot(pending);
        if (pending) { q_id[wave][qn + mbcnt64(pm)] = (uint16_t)((lane << 3) | (uint32_t)j); pend |= 1u << j; }
        qn += (uint32_t)__popcll(pm);
        hist = (hist << 1) | ((c0 >> (7 - j)) & 1u);
    }
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = 0; base < qn; base += 64u) {
        const uint32_t e = base + lane;
        const uint32_t id = q_id[wave][min(e, qn - 1u)], j = id & 7u, owner = id >> 3;
        // the owner's history at bit position j: its byte-aligned history shifted by the first j bits of its byte
        const uint32_t olo = (uint32_t)__shfl((int)(uint32_t)hist0, (int)owner, 64), ohi = (uint32_t)__shfl((int)(uint32_t)(hist0 >> 32), (int)owner, 64);
        const uint32_t oc0 = (uint32_t)__shfl((int)c0, (int)owner, 64);
        if (e < qn) {
            const uint64_t h = (((((uint64_t)ohi << 32) | olo)) << j) | (uint64_t)(oc0 >> (8u - j));
            const uint4 sv = a.lut[((uint32_t)h & ((1u << W3_ACHASH_LUT_BITS) - 1u)) * 8u + j];
            // StationaryModel::predict walks the bit positions backwards from j: the r-th coded history bit uses table[(j - 1 - r) & 7]
            uint32_t rot[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t k = (j + 7u - (uint32_t)r) & 7u;
                rot[r] = k == 0 ? p32t[0] : k == 1 ? p32t[1] : k == 2 ? p32t[2] : k == 3 ? p32t[3] : k == 4 ? p32t[4] : k == 5 ? p32t[5] : k == 6 ? p32t[6] : p32t[7];
            }
            ACHashState st; st.x1 = sv.x; st.x2 = sv.y; st.hash = sv.z; st.meta = sv.w;
            st = ac_history_hash_steps(h, a.max_bits, rot, st, W3_ACHASH_LUT_BITS, 64);
            q_key[wave][owner][j] = (uint8_t)(ac_hash_finish(st, a.max_bits) & a.hmask);
        }
    }
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; j++)
        if ((pend >> j) & 1u) out[j >> 2] |= (uint32_t)q_key[wave][lane][j] << (8 * (j & 3));
    if (live) a.keys[g] = make_uint2(out[0], out[1]);
}

// FrozenModel as the leftmost leaf: every p is Counter::new().p() = 32768
__global__ void __launch_bounds__(256) k_fill_half(uint4 *P, uint64_t n) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) P[g] = make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u);
}

}  // namespace w3
