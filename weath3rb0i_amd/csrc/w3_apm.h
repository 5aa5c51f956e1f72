// w3_apm.h — APM stages of the two-phase encoder (gfx950).
//
// The APM ("APM mixers", README.md:10 — a goal of the reference, no code; BUILD-DEFINED here, DESIGN.md §2.4,
// the CPU checker restates it as apm_pp/apm_update) refines the probability p of its input model through a table
// t[row][33] of u16 interpolated over stretch(p):
//     pos = (stretch(p) + 2048) * 32, j = pos >> 12, w = pos & 4095
//     pa  = (t[row][j] * (4096 - w) + t[row][j+1] * w) >> 12,      p' = clamp((p + 3 pa + 2) >> 2, 1, 65535)
//     update: the nearer entry t[row][j + (w >> 11)] moves towards 65535*bit by (delta >> rate) (floor).
//
// In the two-phase encoder the stage's INPUT p is known for every step before the stage runs (the predict
// kernels produced it), and so are the row (a function of the input bits) and the coded bit.  What stays
// serial is only the history of each table entry.  One wavefront owns one table in LDS (256 rows x 33 x u16 =
// 16.5 KiB; four waves share the 8 KiB stretch LUT: two workgroups per CU) and walks its positions in time
// order, 8 positions x 8 bit positions per round:  lane = (k = position in the round, j = bit position).
// The 8 lanes of one position touch 8 different rows (the partial byte c0 has a different length per j), so they
// are conflict free; the 8 positions of a round are committed one after another (LDS executes one wave's
// instructions in order), everything else — loads, OpinionMixer2 over the leaf streams, stretch, interpolation,
// the output store — is done for the 64 steps at once.
//
//   k_apm0<L> : row = c0 (W3_APM_ORDER0).  One wave per block, time order; reads the L leaf streams (mixing them
//               on the fly) or the previous stage's stream, writes the stage's stream.  Coalesced 128-B accesses.
//   k_apm1    : row = c0 | c1 << 8 (W3_APM_ORDER1) = 256 independent order-0 tables keyed by the previous byte.
//               Walks the block's records sorted by c1 (k_partition<1>): each group is a contiguous, time-ordered
//               run that starts from a fresh table.  Jobs = (block, slice) handed out block-major (as in
//               k_rank_sorted) so the in-place 16-byte gathers/scatters of P stay in the Infinity Cache.
#pragma once
#include "w3_predict.h"

namespace w3 {

#define W3_APM_WAVES 4
#define W3_APM_TBL (256 * 33)
#define W3_APM_PF 4   // rounds whose loads are in flight together

struct ApmArgs {
    const uint8_t *in;
    uint64_t n;
    uint32_t block_size, nblocks;
    const uint16_t *src[8];    // k_apm0: L input streams (8 x u16 per input byte); k_apm1: unused
    uint16_t *P;               // the stage's output stream (k_apm1: input as well, in place)
    const int16_t *stretch;    // [4096]
    const uint16_t *squash;    // [4095]
    uint32_t rate;
    const uint2 *rec;          // k_apm1: records sorted by c1 (k_partition<1>)
    const uint32_t *splits;    // k_apm1: [nblocks][W3_SLICES + 1]
    uint32_t *job_counter;     // k_apm1
    uint16_t *dummy;           // [64] sink for the stores of lanes past the block end (keeps every store unconditional)
};

// LDS pointers keep their address space (a generic pointer turns every access into a flat_* instruction).
// Lanes of one wave communicate through the table: ordering comes from the hardware (LDS executes one wave's
// instructions in order) plus a compiler barrier wherever one lane group's write must precede another's read.
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) int16_t lds_i16;
typedef __attribute__((address_space(3))) uint64_t lds_u64;
#define W3_LDS_FENCE() __asm__ volatile("" ::: "memory")

// identity map: t[row][j] = squash((j - 16) * 128)
__device__ __forceinline__ void apm_table_init(lds_u16 *tab, const lds_u16 *s_row, int lane) {
    lds_u32 *t32 = (lds_u32 *)tab;
    W3_LDS_FENCE();
    for (uint32_t i = (uint32_t)lane; i < W3_APM_TBL / 2u; i += 64u) {
        const uint32_t e = 2u * i;
        t32[i] = (uint32_t)s_row[e % 33u] | ((uint32_t)s_row[(e + 1u) % 33u] << 16);
    }
    W3_LDS_FENCE();
}

// lane i receives the value of lane i - 8 / i + 8 of its 16-lane DPP row (lanes without a source get 0)
__device__ __forceinline__ uint32_t dpp_from_lane_minus8(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t dpp_from_lane_plus8(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x108 /* row_shl:8 */, 0xF, 0xF, true);
}

// One round: 64 steps = 8 positions (k) x 8 bit positions (j), lane = 8 k + j.  The 8 lanes of one position touch 8
// different rows (no conflicts); the 8 positions are committed in time order, in PAIRS (2m, 2m+1) = one 16-lane DPP row per
// sub-step, under an EXEC mask of that row (an idle lane executes nothing):
//     read the entry to update and its neighbour
//     nv = (old * (2^rate - 1) + 65535 * bit) >> rate          == old + ((65535 * bit - old) >> rate), arithmetic shift
//     the later position takes the earlier one's nv over DPP row_shr:8 where it reads the entry the earlier one updates,
//     and recomputes; the earlier one leaves the store to the later one when both update the same entry
//     write
// Round 1 had every lane execute every sub-step (selects for "is this my pair", dummy slots for idle lanes, the update as
// sub / shift / add): 100 VALU instructions per round in the sub-steps, 138 in all, VALU-bound at 84 %.  A sub-step is now
// 7 VALU + 3 LDS + 4 SALU instructions.  (Committing one position per sub-step needs only 2 VALU each, but 8 serial LDS round
// trips per round instead of 4, and a CU holds only 8 tables = 8 chains: 35.7 ms against 29.7, measured.)
// The round is split in three so that a batch's LUT look-ups and address arithmetic run together, ahead of its sub-steps.
struct ApmPrep {
    uint32_t a_upd, a_oth, tgt, w, hi, p;
    uint64_t vm, fu, fo, sh;
};

__device__ __forceinline__ ApmPrep apm_prep(lds_u16 *tab, const lds_i16 *s_str, uint32_t p, uint32_t row, uint32_t bit, bool valid) {
    ApmPrep q;
    const uint32_t pos = (uint32_t)((int)s_str[p >> 4] + 2048) * 32u;
    q.p = p;
    q.w = pos & 4095u; q.hi = q.w >> 11;
    const uint32_t e = row * 33u + (pos >> 12);
    // LDS byte addresses of the entry this step updates (the nearer one) and of the other one of its pair
    q.a_upd = (uint32_t)(uintptr_t)(tab + e + q.hi); q.a_oth = (uint32_t)(uintptr_t)(tab + e + 1u - q.hi);
    q.tgt = bit ? 65535u : 0u;
    // hazards inside a pair: lanes 0-7 of a DPP row are the earlier position (E), lanes 8-15 the later one (L)
    const bool is_l = (threadIdx.x & 8u) != 0u;
    const uint32_t e_upd = dpp_from_lane_minus8(q.a_upd);                       // L lanes: the entry E updates
    const uint32_t l_upd = dpp_from_lane_plus8(q.a_upd);                        // E lanes: the entry L updates
    const uint32_t l_valid = dpp_from_lane_plus8(valid ? 1u : 0u);
    q.vm = __ballot(valid);                                                     // lanes past the block end take no part
    q.fu = __ballot(valid && is_l && e_upd == q.a_upd);                         // L updates the entry E updates: start from E's new value
    q.fo = __ballot(valid && is_l && e_upd == q.a_oth);                         // L's other entry is the one E updates
    q.sh = __ballot(valid && !is_l && l_valid != 0u && l_upd == q.a_upd);       // E leaves the store to L
    return q;
}

__device__ __forceinline__ void apm_commit(const ApmPrep &q, uint32_t rate, uint32_t &au, uint32_t &ao) {
    const uint32_t mult = (uint32_t)__builtin_amdgcn_readfirstlane((int)((1u << rate) - 1u));
    const uint32_t rate_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)rate);
    uint32_t tmp, fwd;
    uint64_t save, m;
#define W3_APM_SUBSTEP                                            \
    "s_and_b64 exec, %[m], %[vm]\n"                               \
    "ds_read_u16 %[au], %[aupd]\n"                                \
    "ds_read_u16 %[ao], %[aoth]\n"                                \
    "s_waitcnt lgkmcnt(0)\n"                                      \
    "v_mad_u32_u24 %[t], %[au], %[mult], %[tgt]\n"                \
    "v_lshrrev_b32 %[t], %[rate], %[t]\n"                         \
    "s_nop 1\n"                                                   \
    "v_mov_b32_dpp %[f], %[t] row_shr:8 row_mask:0xf bank_mask:0xf\n" \
    "v_cndmask_b32 %[au], %[au], %[f], %[fu]\n"                   \
    "v_cndmask_b32 %[ao], %[ao], %[f], %[fo]\n"                   \
    "v_mad_u32_u24 %[t], %[au], %[mult], %[tgt]\n"                \
    "v_lshrrev_b32 %[t], %[rate], %[t]\n"                         \
    "s_andn2_b64 exec, exec, %[sh]\n"                             \
    "ds_write_b16 %[aupd], %[t]\n"                                \
    "s_lshl_b64 %[m], %[m], 16\n"
    asm volatile("s_mov_b64 %[save], exec\n"
                 "s_mov_b64 %[m], 0xffff\n"
                 "v_mov_b32 %[f], 0\n"
                 W3_APM_SUBSTEP W3_APM_SUBSTEP W3_APM_SUBSTEP W3_APM_SUBSTEP
                 "s_mov_b64 exec, %[save]\n"
                 : [au] "=&v"(au), [ao] "=&v"(ao), [t] "=&v"(tmp), [f] "=&v"(fwd), [save] "=&s"(save), [m] "=&s"(m)
                 : [aupd] "v"(q.a_upd), [aoth] "v"(q.a_oth), [tgt] "v"(q.tgt), [mult] "s"(mult), [rate] "s"(rate_s), [vm] "s"(q.vm),
                   [fu] "s"(q.fu), [fo] "s"(q.fo), [sh] "s"(q.sh)
                 : "memory", "scc");
#undef W3_APM_SUBSTEP
}

__device__ __forceinline__ uint32_t apm_finish(const ApmPrep &q, uint32_t au, uint32_t ao) {
    const uint32_t t0 = q.hi ? ao : au, t1 = q.hi ? au : ao;
    const uint32_t pa = (t0 * (4096u - q.w) + t1 * q.w) >> 12;
    const uint32_t o = (q.p + 3u * pa + 2u) >> 2;
    return o < 1u ? 1u : o > 65535u ? 65535u : o;
}

// the three stages of one round in one call (k_apm1)
__device__ __forceinline__ uint32_t apm_round(lds_u16 *tab, const lds_i16 *s_str, uint32_t p, uint32_t row, uint32_t bit,
                                              bool valid, uint32_t rate, int k) {
    (void)k;
    const ApmPrep q = apm_prep(tab, s_str, p, row, bit, valid);
    uint32_t au, ao;
    apm_commit(q, rate, au, ao);
    return apm_finish(q, au, ao);
}

// k_apm0: TWO wavefronts per block take alternate batches of W3_APM_PF rounds.  Only the sub-steps (stage 2) touch the
// table, and they are a latency chain (an LDS round trip per pair of positions) that leaves the VALU mostly idle, while the
// rest of a round (loads, mix, LUT look-ups, interpolation, store: stages 1 and 3) needs no table at all.  With one wave per
// block the table sat unused for ~45 % of the time and a CU holds only 8 tables; with two, one wave prepares its next batch
// while the other commits, handing the table over through one LDS word per block (the batch whose turn it is).
template <int L>
__global__ void __launch_bounds__(128 * W3_APM_WAVES) k_apm0(ApmArgs a) {
    __shared__ uint16_t s_tab[W3_APM_WAVES][W3_APM_TBL + 128];
    __shared__ int16_t s_str[4096];
    __shared__ uint16_t s_row[34];
    __shared__ uint32_t s_turn[W3_APM_WAVES];   // per block: the batch that may commit next
    for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) s_str[i] = a.stretch[i];
    if (threadIdx.x < 33u) {
        int d = ((int)threadIdx.x - 16) * 128;
        d = d < -2047 ? -2047 : d > 2047 ? 2047 : d;
        s_row[threadIdx.x] = a.squash[d + 2047];
    }
    if (threadIdx.x < W3_APM_WAVES) s_turn[threadIdx.x] = 0u;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, k = lane >> 3, j = lane & 7;
    const int slot = wave >> 1;                 // block (table) of this wave inside the workgroup
    const uint32_t role = (uint32_t)wave & 1u;  // which batches: role, role + 2, ...
    lds_u16 *tab = (lds_u16 *)&s_tab[slot][0];
    const lds_i16 *l_str = (const lds_i16 *)&s_str[0];
    const lds_u16 *l_row = (const lds_u16 *)&s_row[0];
    volatile uint32_t *turn = &s_turn[slot];
    // wave-uniform block id in an SGPR: the per-leaf base addresses become scalar and every access is base + 32-bit lane offset
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * W3_APM_WAVES + (uint32_t)slot));
    if (b >= a.nblocks) return;   // (no barrier below)
    const uint64_t off = (uint64_t)b * a.block_size;
    const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
    const uint32_t last = len - 1u;
    const uint8_t *blk = a.in + off;
    uint16_t *out = a.P + off * 8u;
    constexpr uint32_t BATCH = 8u * W3_APM_PF;   // positions per batch
    const uint32_t nbatch = (len + BATCH - 1u) / BATCH;
    // Operands of a batch are loaded one (own) batch ahead, unconditionally (index clamped; see k_coder_fast), into
    // two register sets used alternately: rotating one set through copies at the loop top made hipcc wait for the
    // previous batch's STORES (vmcnt counts loads and stores in one queue) before every copy.
    uint32_t pA[W3_APM_PF][L], bA[W3_APM_PF], pB[W3_APM_PF][L], bB[W3_APM_PF];
    auto load = [&](uint32_t (&pp)[W3_APM_PF][L], uint32_t (&bb)[W3_APM_PF], uint32_t base) {
#pragma unroll
        for (int r = 0; r < W3_APM_PF; r++) {
            const uint32_t ic = min(base + (uint32_t)(r * 8 + k), last);
            bb[r] = blk[ic];
#pragma unroll
            for (int l = 0; l < L; l++)   // scalar base + 32-bit byte offset (global_load ... v_off, s[base])
                pp[r][l] = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(a.src[l] + off * 8u) + (ic * 16u + 2u * (uint32_t)j));
        }
    };
    auto process = [&](const uint32_t (&pc)[W3_APM_PF][L], const uint32_t (&bc)[W3_APM_PF], uint32_t bi) {
        const uint32_t base = bi * BATCH;
        ApmPrep q[W3_APM_PF];
        uint32_t au[W3_APM_PF], ao[W3_APM_PF];
#pragma unroll
        for (int r = 0; r < W3_APM_PF; r++) {   // stage 1 of the batch: mix, LUT look-ups, addresses, hazard masks (no table access)
            const uint32_t i = base + (uint32_t)(r * 8 + k);
            const bool valid = i < len;
            uint32_t p = pc[r][0];
            if constexpr (L > 1) {   // OpinionMixer2 over the leaves: leftmost of maximal |p - 1/2| (models/mod.rs:67-69)
                uint32_t d = opinion_dist(p);
#pragma unroll
                for (int l = 1; l < L; l++) {
                    const uint32_t qq = pc[r][l], dq = opinion_dist(qq);
                    if (dq > d) { p = qq; d = dq; }
                }
            }
            const uint32_t byte = bc[r];
            const uint32_t c0 = (1u << j) | (byte >> (8 - j));        // partial byte with a leading 1
            const uint32_t bit = (byte >> (7 - j)) & 1u;
            q[r] = apm_prep(tab, l_str, p, c0, bit, valid);
        }
        // stage 2, the serial part: wait for this batch's turn on the block's table
        if (bi == 0u) apm_table_init(tab, l_row, lane);
        else {
            while (__hip_atomic_load(const_cast<const uint32_t *>(turn), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < bi) __builtin_amdgcn_s_sleep(1);
        }
        W3_LDS_FENCE();
#pragma unroll
        for (int r = 0; r < W3_APM_PF; r++) apm_commit(q[r], a.rate, au[r], ao[r]);   // (rounds past the end: EXEC empty)
        W3_LDS_FENCE();
        __hip_atomic_store(const_cast<uint32_t *>(turn), bi + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int r = 0; r < W3_APM_PF; r++) {   // stage 3: interpolate, store
            const uint32_t i = base + (uint32_t)(r * 8 + k);
            const bool valid = i < len;
            const uint32_t o = apm_finish(q[r], au[r], ao[r]);
            // unconditional store (a branch around it makes hipcc wait vmcnt(0) — store latency included — before it
            // touches the prefetched operands of the next batch)
            uint16_t *dst = valid ? reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(out) + (i * 16u + 2u * (uint32_t)j)) : a.dummy + lane;
            *dst = (uint16_t)o;
        }
    };
    load(pA, bA, role * BATCH);
    for (uint32_t bi = role; bi < nbatch; bi += 4u) {
        load(pB, bB, (bi + 2u) * BATCH);
        process(pA, bA, bi);
        if (bi + 2u >= nbatch) break;
        load(pA, bA, (bi + 4u) * BATCH);
        process(pB, bB, bi + 2u);
    }
}

__global__ void __launch_bounds__(64 * W3_APM_WAVES) k_apm1(ApmArgs a) {
    __shared__ uint16_t s_tab[W3_APM_WAVES][W3_APM_TBL + 128];   // + 64 dummy u16 pairs (apm_round)
    __shared__ int16_t s_str[4096];
    __shared__ uint16_t s_row[34];
    for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) s_str[i] = a.stretch[i];
    if (threadIdx.x < 33u) {
        int d = ((int)threadIdx.x - 16) * 128;
        d = d < -2047 ? -2047 : d > 2047 ? 2047 : d;
        s_row[threadIdx.x] = a.squash[d + 2047];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, k = lane >> 3, j = lane & 7;
    lds_u16 *tab = (lds_u16 *)&s_tab[wave][0];
    const lds_i16 *l_str = (const lds_i16 *)&s_str[0];
    const lds_u16 *l_row = (const lds_u16 *)&s_row[0];
    const uint32_t njobs = a.nblocks * W3_SLICES;
    for (;;) {
        uint32_t job = 0;
        if (lane == 0) job = atomicAdd(a.job_counter, 1u);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) break;
        const uint32_t b = job / W3_SLICES, sl = job % W3_SLICES;
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint32_t *sp = a.splits + (uint64_t)b * (W3_SLICES + 1u);
        const uint32_t lo = sp[sl], hi_e = sp[sl + 1];
        if (lo >= hi_e) continue;
        const uint32_t len = hi_e - lo, last = len - 1u;
        const uint2 *rec = a.rec + off + lo;
        uint16_t *P = a.P + off * 8u;
        uint32_t open_g = 0xFFFFFFFFu;   // the group the table describes (none yet)
        // two-level pipeline: records two batches ahead, the P gathers they address one batch ahead
        uint2 rn[W3_APM_PF], rnn[W3_APM_PF];
        uint32_t pn[W3_APM_PF];
#pragma unroll
        for (int r = 0; r < W3_APM_PF; r++) {
            rn[r] = rec[min((uint32_t)(r * 8 + k), last)];
            rnn[r] = rec[min((uint32_t)((W3_APM_PF + r) * 8 + k), last)];
        }
#pragma unroll
        for (int r = 0; r < W3_APM_PF; r++) pn[r] = P[(uint64_t)rn[r].x * 8u + (uint32_t)j];
        for (uint32_t base = 0; base < len; base += 8u * W3_APM_PF) {
            uint2 rc[W3_APM_PF];
            uint32_t pc[W3_APM_PF];
#pragma unroll
            for (int r = 0; r < W3_APM_PF; r++) { rc[r] = rn[r]; pc[r] = pn[r]; rn[r] = rnn[r]; }
#pragma unroll
            for (int r = 0; r < W3_APM_PF; r++) rnn[r] = rec[min(base + (uint32_t)((2 * W3_APM_PF + r) * 8 + k), last)];
#pragma unroll
            for (int r = 0; r < W3_APM_PF; r++) pn[r] = P[(uint64_t)rn[r].x * 8u + (uint32_t)j];
#pragma unroll
            for (int r = 0; r < W3_APM_PF; r++) {
                if (base + (uint32_t)(r * 8) >= len) break;
                const uint32_t e = base + (uint32_t)(r * 8 + k);
                const bool valid = e < len;
                const uint32_t wv = rc[r].y, byte = wv & 0xFFu, g = (wv >> 8) & 0xFFu;
                const uint32_t c0 = (1u << j) | (byte >> (8 - j));
                const uint32_t bit = (byte >> (7 - j)) & 1u;
                const uint32_t p = pc[r];
                // groups (previous byte c1) are contiguous and time ordered; each starts from a fresh table
                const uint64_t vm = __ballot(valid);
                const uint64_t same = __ballot(valid && g == open_g);
                uint32_t o;
                if (same == vm) {
                    o = apm_round(tab, l_str, p, c0, bit, valid, a.rate, k);
                } else {
                    // a group boundary inside the round: commit position by position, re-initialising between groups
                    const uint32_t pos = (uint32_t)((int)l_str[p >> 4] + 2048) * 32u;
                    const uint32_t w = pos & 4095u, hi = w >> 11;
                    const uint32_t ent = c0 * 33u + (pos >> 12);
                    const int target = bit ? 65535 : 0;
                    uint32_t t0 = 0u, t1 = 0u;
#pragma unroll 1
                    for (int kk = 0; kk < 8; kk++) {
                        if (!((vm >> (kk * 8)) & 1ull)) break;
                        const uint32_t gk = readlane_u32(g, kk * 8);
                        if (gk != open_g) { apm_table_init(tab, l_row, lane); open_g = gk; }
                        if (k == kk) {
                            t0 = tab[ent]; t1 = tab[ent + 1u];
                            const int tv = (int)(hi ? t1 : t0);
                            tab[ent + hi] = (uint16_t)(tv + ((target - tv) >> a.rate));
                        }
                        W3_LDS_FENCE();
                    }
                    const uint32_t pa = (t0 * (4096u - w) + t1 * w) >> 12;
                    o = (p + 3u * pa + 2u) >> 2;
                    o = o < 1u ? 1u : o > 65535u ? 65535u : o;
                }
                uint16_t *dst = valid ? P + ((uint64_t)rc[r].x * 8u + (uint32_t)j) : a.dummy + lane;
                *dst = (uint16_t)o;
            }
        }
    }
}

}  // namespace w3
