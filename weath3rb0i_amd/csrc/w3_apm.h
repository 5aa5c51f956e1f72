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
// serial is only the history of each table entry.  A block's table lives in LDS (256 rows x 33 x u16 = 16.5 KiB; the
// four tables of a workgroup share one 8 KiB LUT: two workgroups = eight tables per CU) and its positions are walked in
// time order, 8 positions x 8 bit positions per round:  lane = (k = position in the round, j = bit position).
// The 8 lanes of one position touch 8 different rows (the partial byte c0 has a different length per j), so they
// are conflict free; the 8 positions of a round are committed one after another (LDS executes one wave's
// instructions in order), everything else — loads, OpinionMixer2 over the leaf streams, stretch, interpolation,
// the output store — is done for the 64 steps at once.
//
//   k_apm0<L> : row = c0 (W3_APM_ORDER0).  Two waves per block take alternate batches of 4 rounds, time order; reads the L
//               leaf streams (mixing them on the fly) or the previous stage's stream, writes the stage's stream, 8 bytes per lane.
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
    uint16_t *dummy;           // 2 KiB sink for the stores of lanes past the block end (keeps every store unconditional)
    uint32_t *oob;             // -DW3_TUNING builds: counts stores whose address lies neither in [P, P + 16 n) nor in the sink (else null)
};

// Debug-build guard of every global store of the APM kernels ("the round-2 faults"): the address must lie
// inside the stage's output stream or inside the sink.  A store that does not is counted and redirected to the sink, and the
// host turns the count into W3_E_HIP — instead of a memory access fault that may take the GPU down.
#ifdef W3_TUNING
#define W3_APM_CHECK_STORE(a, ptr, bytes)                                                                     \
    do {                                                                                                      \
        const uintptr_t p_ = (uintptr_t)(ptr), lo_ = (uintptr_t)(a).P, hi_ = lo_ + (uintptr_t)(a).n * 16u;    \
        const uintptr_t s_ = (uintptr_t)(a).dummy;                                                            \
        if (!((p_ >= lo_ && p_ + (bytes) <= hi_) || (p_ >= s_ && p_ + (bytes) <= s_ + 2048u))) {              \
            if ((a).oob) atomicAdd((a).oob, 1u);                                                              \
            ptr = reinterpret_cast<decltype(ptr)>((a).dummy);                                                 \
        }                                                                                                     \
    } while (0)
#else
#define W3_APM_CHECK_STORE(a, ptr, bytes) do { } while (0)
#endif

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

// lane i receives the value of lane i - 8 of its 16-lane DPP row (lanes without a source get 0)
__device__ __forceinline__ uint32_t dpp_from_lane_minus8(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
}

// One round: 64 steps = 8 positions (k) x 8 bit positions (j), lane = 8 k + j.  The 8 lanes of one position touch 8 different
// rows (no conflicts); the 8 positions are committed in time order, in PAIRS (2m, 2m+1) = one 16-lane DPP row per sub-step,
// under an EXEC mask of that row (an idle lane executes nothing):
//     read the entry to update and its neighbour
//     nv = (old * (2^rate - 1) + 65535 * bit) >> rate          == old + ((65535 * bit - old) >> rate), arithmetic shift
//     the later position takes the earlier one's nv (DPP row_shr:8) where it reads the entry the earlier one updates, and
//     recomputes; the earlier one leaves the store to the later one when both update the same entry
//     write
// Round 1 had every lane execute every sub-step (selects for "is this my pair", dummy slots for idle lanes, the update as
// sub / shift / add): 100 VALU instructions per round in the sub-steps, 138 in all, VALU-bound at 84 %.  A sub-step is now
// 6 VALU + 3 LDS + 5 SALU instructions.  (Committing one position per sub-step needs only 2 VALU each, but 8 serial LDS round
// trips per round instead of 4, and a CU holds only 8 tables = 8 chains: 35.7 ms against 29.7, measured.)
// The round is split in three (apm0_prep / apm0_commit / apm0_finish below) so that a batch's LUT look-ups and address
// arithmetic run together, ahead of its sub-steps.

// ---------------------------------------------------------------------------
// k_apm0, second form (round 2).  TWO wavefronts per block take alternate batches of W3_APM_PF rounds: only the sub-steps
// (stage 2) touch the table, and they are a latency chain (an LDS round trip per pair of positions) that leaves the VALU mostly
// idle, while the rest of a round (loads, mix, LUT look-ups, interpolation, store: stages 1 and 3) needs no table at all.
// With one wave per block the table sat unused for ~45 % of the time and a CU holds only 8 tables; with two, one wave
// prepares its next batch while the other commits, handing the table over through one LDS word per block.
// With 16 waves per CU the kernel is bound by the VALU instruction count again, so everything around the sub-steps is
// written for few instructions:
//   * a batch's 32 positions x 16 B of every input stream arrive as ONE 8-byte load per lane (lane = position, half),
//     OpinionMixer2 runs on that layout with packed 16-bit VALU (two steps per instruction), and only the mixed stream is
//     turned into the round layout (lane = position in round, bit position) through 512 B of LDS per wave; the output goes
//     the same way back and leaves as one 8-byte store per lane.  (Before: 2-byte loads and stores, 20 memory instructions
//     and their address arithmetic per batch instead of L + 1.)
//   * the batch's 32 input bytes come as one dword load (lane & 7 = dword); a lane takes its bytes with ds_bpermute + a shift
//   * the LUT holds, per stretch(p) bucket, the fields a step needs, ready made: 2 * (j + hi) | x << 7 | (1 - hi) << 14 with
//     x = the interpolation weight of the entry that is NOT updated, in 1/128:  pa = au + (((ao - au) * x) >> 7)  — equal to
//     (t[j] * (4096 - w) + t[j+1] * w) >> 12 because w is a multiple of 32
//   * hazard masks of a pair from one DPP move and two compares; in the sub-step the forwarded value is taken by v_cndmask with a DPP source
// ---------------------------------------------------------------------------
#define W3_APM0_LUT_B   0u                                      // [4096] u16 (first: ds offsets of the look-ups fold into the instruction)
#define W3_APM0_TAB_B   8192u                                   // W3_APM_WAVES tables of W3_APM_TBL u16
#define W3_APM0_STAGE_B (W3_APM0_TAB_B + W3_APM_WAVES * W3_APM_TBL * 2u)   // 2 * W3_APM_WAVES waves x 512 B
#define W3_APM0_ROW_B   (W3_APM0_STAGE_B + 2u * W3_APM_WAVES * 512u)      // [33] u16 (identity row), padded to 80 B
#define W3_APM0_TURN_B  (W3_APM0_ROW_B + 80u)                   // [W3_APM_WAVES] u32
#define W3_APM0_LDS     (W3_APM0_TURN_B + 4u * W3_APM_WAVES)

// LUT entry of one stretch(p) bucket: (stretch + 2048) * 32 = j << 12 | w;  + 64 rounds to the nearer entry: t >> 7 = j + hi
__device__ __forceinline__ uint16_t apm_lut_entry(int stretch) {
    const uint32_t t = (uint32_t)(stretch + 2048 + 64);
    const uint32_t u = t & 127u, x = u < 64u ? 64u - u : u - 64u;    // w >> 5 = u ^ 64;  x = weight (in 1/128) of the entry that is not updated
    return (uint16_t)(((t >> 7) << 1) | (x << 7) | (((t >> 6) & 1u) << 15));   // bit 15 = 1 - hi: the other entry is the one above
}

struct Apm0Prep {
    uint32_t X;       // LDS byte address of the entry this step updates, MINUS 2 (the ds instructions carry offset:2)
    uint32_t a_oth;   // LDS byte address of the other entry of its pair
    uint32_t tgt, x, p2;
    uint64_t nfu, nfo, sh;   // ~(L takes E's new value as the entry it updates / as its other entry); E leaves the store to L
};

// one step's stage 1.  `byte` may carry garbage above bit 7.  lm / em: later / earlier lanes of every pair that take part.
__device__ __forceinline__ Apm0Prep apm0_prep(uint32_t tabm2, uint32_t lds0, uint32_t p, uint32_t byte, uint32_t j, uint32_t onej, uint64_t lm, uint64_t em) {
    Apm0Prep q;
    const uint32_t lut = *(const lds_u16 *)(uintptr_t)(lds0 + W3_APM0_LUT_B + ((p >> 3) & 0x1FFEu));
    const uint32_t c0 = __builtin_amdgcn_ubfe(byte, 8u - j, j) | onej;        // partial byte with a leading 1 = the table row
    q.X = __umul24(c0, 66u) + ((lut & 0x7Eu) + tabm2);
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(q.a_oth) : "v"(lut >> 15), "v"(q.X));   // X + 4 * (1 - hi); as asm: hipcc turns the C form into shift / and / add
    const uint32_t om2 = q.a_oth - 2u;
    q.x = __builtin_amdgcn_ubfe(lut, 7u, 7u);
    q.tgt = (uint32_t)__builtin_amdgcn_sbfe((int)byte, 7u - j, 1u) & 0xFFFFu;
    q.p2 = p + 2u;
    // lanes 0-7 of a DPP row are the earlier position (E), lanes 8-15 the later one (L).  (gfx950 has no DPP form of v_cmp.)
    const uint32_t e_x = dpp_from_lane_minus8(q.X);                 // L lanes: E's X
    const uint64_t m_u = __ballot(e_x == q.X);                      // L: the entry E updates is the one I update
    const uint64_t m_o = __ballot(e_x == om2);                      // L: the entry E updates is my other one
    q.nfu = ~(m_u & lm); q.nfo = ~(m_o & lm);                       // 1 = keep the value read from the table
    q.sh = (m_u >> 8) & em;                                         // E: L updates the entry I update, and stores for both
    return q;
}

__device__ __forceinline__ void apm0_commit(const Apm0Prep &q, uint32_t mult, uint32_t rate_s, uint64_t vm, uint32_t &au, uint32_t &ao) {
    uint32_t tmp;
    uint64_t save, m;
#define W3_APM0_SUBSTEP                                           \
    "s_and_b64 exec, %[m], %[vm]\n"                               \
    "ds_read_u16 %[au], %[X] offset:2\n"                          \
    "ds_read_u16 %[ao], %[aoth]\n"                                \
    "s_mov_b64 vcc, %[nfu]\n"                                     \
    "s_waitcnt lgkmcnt(0)\n"                                      \
    "v_mad_u32_u24 %[t], %[au], %[mult], %[tgt]\n"                \
    "v_lshrrev_b32 %[t], %[rate], %[t]\n"                         \
    "s_nop 1\n"                                                   \
    "v_cndmask_b32_dpp %[au], %[t], %[au], vcc row_shr:8 row_mask:0xf bank_mask:0xf\n" \
    "s_mov_b64 vcc, %[nfo]\n"                                     \
    "v_cndmask_b32_dpp %[ao], %[t], %[ao], vcc row_shr:8 row_mask:0xf bank_mask:0xf\n" \
    "v_mad_u32_u24 %[t], %[au], %[mult], %[tgt]\n"                \
    "v_lshrrev_b32 %[t], %[rate], %[t]\n"                         \
    "s_andn2_b64 exec, exec, %[sh]\n"                             \
    "ds_write_b16 %[X], %[t] offset:2\n"                          \
    "s_lshl_b64 %[m], %[m], 16\n"
    asm volatile("s_mov_b64 %[save], exec\n"
                 "s_mov_b64 %[m], 0xffff\n"
                 W3_APM0_SUBSTEP W3_APM0_SUBSTEP W3_APM0_SUBSTEP W3_APM0_SUBSTEP
                 "s_mov_b64 exec, %[save]\n"
                 : [au] "=&v"(au), [ao] "=&v"(ao), [t] "=&v"(tmp), [save] "=&s"(save), [m] "=&s"(m)
                 : [X] "v"(q.X), [aoth] "v"(q.a_oth), [tgt] "v"(q.tgt), [mult] "s"(mult), [rate] "s"(rate_s), [vm] "s"(vm),
                   [nfu] "s"(q.nfu), [nfo] "s"(q.nfo), [sh] "s"(q.sh)
                 : "memory", "scc", "vcc");
#undef W3_APM0_SUBSTEP
}

__device__ __forceinline__ uint32_t apm0_finish(const Apm0Prep &q, uint32_t au, uint32_t ao) {
    const int d = (int)ao - (int)au;
    const uint32_t pa = au + (uint32_t)(__mul24(d, (int)q.x) >> 7);
    const uint32_t o = (__umul24(pa, 3u) + q.p2) >> 2;          // <= 65535: p and pa are
    return o < 1u ? 1u : o;
}

struct Apm0Ctx {   // what a wave knows about its block (wave-uniform values in SGPRs)
    uint32_t lds0, tabm2, stage, lane, k, j, onej, bp0, sh8k, len, last, nbatch, role, mult, rate_s;
    lds_u16 *tab; const lds_u16 *l_row; volatile uint32_t *turn;
    const uint8_t *blk; uint8_t *out; uint64_t off;
};

// FAST: the block is a whole number of batches long (every block but a ragged last one, when the block size is a multiple
// of 32): one dword load per batch for the input bytes, constant lane masks, plain stores.
template <int L, bool FAST>
__device__ __forceinline__ void apm0_block(const ApmArgs &a, const Apm0Ctx &c) {
    constexpr uint32_t BATCH = 8u * W3_APM_PF;   // positions per batch
    static_assert(BATCH == 32u, "one 8-byte load per lane covers a batch");
    const uint32_t lane = c.lane, k = c.k;
    // Operands of a batch are loaded one (own) batch ahead, unconditionally (index clamped), into two register sets used
    // alternately (rotating one set through copies made hipcc wait for the previous batch's STORES before every copy).
    uint2 pA[L], pB[L];
    uint32_t yA = 0u, yB = 0u;            // FAST: dword (lane & 7) of the batch's 32 input bytes
    uint32_t sA[W3_APM_PF], sB[W3_APM_PF];   // !FAST: one input byte per round and lane
    auto load = [&](uint2 (&pp)[L], uint32_t &yy, uint32_t (&ss)[W3_APM_PF], uint32_t base) {
        const uint32_t ic = min(base + (lane >> 1), c.last);   // (the prefetch runs up to two batches past the block's end)
        const uint32_t bo = ic * 16u + (lane & 1u) * 8u;
#pragma unroll
        for (int l = 0; l < L; l++)   // scalar base + 32-bit byte offset (global_load_dwordx2 v, v_off, s[base])
            pp[l] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(a.src[l] + c.off * 8u) + bo);
        if constexpr (FAST) {
            __builtin_memcpy(&yy, c.blk + min(base + 4u * (lane & 7u), c.len - 4u), 4);   // (unaligned dword loads are fine on gfx9)
        } else {
#pragma unroll
            for (int r = 0; r < W3_APM_PF; r++) ss[r] = c.blk[min(base + (uint32_t)(r * 8) + k, c.last)];
        }
    };
    auto process = [&](const uint2 (&pc)[L], uint32_t yc, const uint32_t (&sc)[W3_APM_PF], uint32_t bi) {
        const uint32_t base = bi * BATCH;
        // ---- stage 1: mix (load layout), turn into the round layout, look-ups, addresses, hazard masks; no table access
        uint32_t w0 = pc[0].x, w1 = pc[0].y;
        if constexpr (L > 1) {   // OpinionMixer2 over the leaves: leftmost of maximal |p - 1/2| (mixers/opinion_mixer2.rs:5-10), two steps per dword
            uint32_t d0 = as_u32(pk_opinion_dist(as_u16x2(w0))), d1 = as_u32(pk_opinion_dist(as_u16x2(w1)));
#pragma unroll
            for (int l = 1; l < L; l++) {
                const uint32_t q0 = pc[l].x, q1 = pc[l].y;
                const uint32_t e0 = as_u32(pk_opinion_dist(as_u16x2(q0))), e1 = as_u32(pk_opinion_dist(as_u16x2(q1)));
                const uint32_t m0 = pk_farther_mask(as_u16x2(d0), as_u16x2(e0)), m1 = pk_farther_mask(as_u16x2(d1), as_u16x2(e1));   // 0xFFFF where the new leaf is farther from 1/2
                w0 = (q0 & m0) | (w0 & ~m0); w1 = (q1 & m1) | (w1 & ~m1);
                d0 = as_u32(__builtin_elementwise_max(as_u16x2(d0), as_u16x2(e0))); d1 = as_u32(__builtin_elementwise_max(as_u16x2(d1), as_u16x2(e1)));
            }
        }
        W3_LDS_FENCE();
        *(lds_u64 *)(uintptr_t)(c.stage + 8u * lane) = (uint64_t)w0 | ((uint64_t)w1 << 32);
        W3_LDS_FENCE();
        uint32_t p[W3_APM_PF];
#pragma unroll
        for (int r = 0; r < W3_APM_PF; r++) p[r] = *(const lds_u16 *)(uintptr_t)(c.stage + 2u * lane + 128u * (uint32_t)r);
        W3_LDS_FENCE();
        Apm0Prep q[W3_APM_PF];
        uint64_t vmr[W3_APM_PF];
        uint32_t au[W3_APM_PF], ao[W3_APM_PF];
#pragma unroll
        for (int r = 0; r < W3_APM_PF; r++) {
            uint64_t vm = ~0ull;
            uint32_t byte;
            if constexpr (FAST) {
                byte = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c.bp0 + 8u * (uint32_t)r), (int)yc) >> c.sh8k;   // byte 8 r + k of the batch
            } else {
                const uint32_t rb = base + (uint32_t)(r * 8);
                const uint32_t nval = rb >= c.len ? 0u : min(c.len - rb, 8u);             // positions of this round inside the block (scalar)
                vm = nval >= 8u ? ~0ull : ((1ull << (8u * nval)) - 1ull);
                byte = sc[r];
            }
            vmr[r] = vm;
            const uint64_t lm = vm & 0xFF00FF00FF00FF00ull, em = vm & (vm >> 8) & 0x00FF00FF00FF00FFull;
            q[r] = apm0_prep(c.tabm2, c.lds0, p[r], byte, c.j, c.onej, lm, em);
        }
        // ---- stage 2, the serial part: wait for this batch's turn on the block's table
        if (bi == 0u) apm_table_init(c.tab, c.l_row, (int)lane);
        else {
            while (__hip_atomic_load(const_cast<const uint32_t *>(c.turn), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < bi) __builtin_amdgcn_s_sleep(1);
        }
        W3_LDS_FENCE();
        __builtin_amdgcn_s_setprio(3);   // the chain sets the pace: its instructions go ahead of the other waves' stage 1 / 3 work on this SIMD
#pragma unroll
        for (int r = 0; r < W3_APM_PF; r++) apm0_commit(q[r], c.mult, c.rate_s, vmr[r], au[r], ao[r]);   // (rounds past the end: EXEC empty)
        __builtin_amdgcn_s_setprio(0);
        W3_LDS_FENCE();
        __hip_atomic_store(const_cast<uint32_t *>(c.turn), bi + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        // ---- stage 3: interpolate, back to the load layout, store
#pragma unroll
        for (int r = 0; r < W3_APM_PF; r++)
            *(lds_u16 *)(uintptr_t)(c.stage + 2u * lane + 128u * (uint32_t)r) = (uint16_t)apm0_finish(q[r], au[r], ao[r]);
        W3_LDS_FENCE();
        const uint64_t ov = *(const lds_u64 *)(uintptr_t)(c.stage + 8u * lane);
        W3_LDS_FENCE();
        const uint32_t ip = base + (lane >> 1);
        uint2 *dst = reinterpret_cast<uint2 *>(c.out + (ip * 16u + (lane & 1u) * 8u));
        // unconditional store (a branch around it makes hipcc wait vmcnt(0) — store latency included — before it touches the
        // prefetched operands of the next batch); lanes past the block end write to the sink
        if constexpr (!FAST) dst = ip < c.len ? dst : reinterpret_cast<uint2 *>(a.dummy) + lane;
        W3_APM_CHECK_STORE(a, dst, 8u);
        *dst = make_uint2((uint32_t)ov, (uint32_t)(ov >> 32));
    };
    load(pA, yA, sA, c.role * BATCH);
    for (uint32_t bi = c.role; bi < c.nbatch; bi += 4u) {
        load(pB, yB, sB, (bi + 2u) * BATCH);
        process(pA, yA, sA, bi);
        if (bi + 2u >= c.nbatch) break;
        load(pA, yA, sA, (bi + 4u) * BATCH);
        process(pB, yB, sB, bi + 2u);
    }
}

template <int L>
__global__ void __launch_bounds__(128 * W3_APM_WAVES) k_apm0(ApmArgs a) {
    __shared__ uint4 s_mem[(W3_APM0_LDS + 15u) / 16u];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u16 *)&s_mem[0];
    {
        lds_u16 *lut = (lds_u16 *)(uintptr_t)(lds0 + W3_APM0_LUT_B);
        for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) {
            lut[i] = apm_lut_entry((int)a.stretch[i]);
        }
        lds_u16 *row = (lds_u16 *)(uintptr_t)(lds0 + W3_APM0_ROW_B);
        if (threadIdx.x < 33u) {
            int d = ((int)threadIdx.x - 16) * 128;
            d = d < -2047 ? -2047 : d > 2047 ? 2047 : d;
            row[threadIdx.x] = a.squash[d + 2047];
        }
        if (threadIdx.x < W3_APM_WAVES) ((lds_u32 *)(uintptr_t)(lds0 + W3_APM0_TURN_B))[threadIdx.x] = 0u;
    }
    __syncthreads();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t slot = wave >> 1;            // block (table) of this wave inside the workgroup
    Apm0Ctx c;
    c.lds0 = lds0;
    c.lane = threadIdx.x & 63u; c.k = c.lane >> 3; c.j = c.lane & 7u; c.onej = 1u << c.j; c.bp0 = 4u * (c.k >> 2); c.sh8k = 8u * (c.k & 3u);
    c.role = wave & 1u;                         // which batches: role, role + 2, ...
    c.tab = (lds_u16 *)(uintptr_t)(lds0 + W3_APM0_TAB_B + slot * (W3_APM_TBL * 2u));
    c.tabm2 = lds0 + W3_APM0_TAB_B + slot * (W3_APM_TBL * 2u) - 2u;
    c.l_row = (const lds_u16 *)(uintptr_t)(lds0 + W3_APM0_ROW_B);
    c.turn = (volatile uint32_t *)&((uint32_t *)&s_mem[0])[W3_APM0_TURN_B / 4u + slot];
    c.stage = lds0 + W3_APM0_STAGE_B + wave * 512u;
    // wave-uniform block id in an SGPR: the per-leaf base addresses become scalar and every access is base + 32-bit lane offset
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * W3_APM_WAVES + slot));
    if (b >= a.nblocks) return;   // (no barrier below)
    c.off = (uint64_t)b * a.block_size;
    c.len = (uint32_t)((a.n - c.off) < a.block_size ? (a.n - c.off) : a.block_size);
    c.last = c.len - 1u;
    c.blk = a.in + c.off;
    c.out = reinterpret_cast<uint8_t *>(a.P + c.off * 8u);
    c.nbatch = (c.len + 31u) / 32u;
    c.mult = (uint32_t)__builtin_amdgcn_readfirstlane((int)((1u << a.rate) - 1u));
    c.rate_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.rate);
    if ((c.len & 31u) == 0u) apm0_block<L, true>(a, c);
    else apm0_block<L, false>(a, c);
}

__global__ void __launch_bounds__(64 * W3_APM_WAVES) k_apm1(ApmArgs a) {
    __shared__ uint16_t s_lut[4096];                              // first: its LDS address is the `lds0` of apm0_prep
    __shared__ uint16_t s_tab[W3_APM_WAVES][W3_APM_TBL + 2];
    __shared__ uint16_t s_row[34];
    for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) s_lut[i] = apm_lut_entry((int)a.stretch[i]);
    if (threadIdx.x < 33u) {
        int d = ((int)threadIdx.x - 16) * 128;
        d = d < -2047 ? -2047 : d > 2047 ? 2047 : d;
        s_row[threadIdx.x] = a.squash[d + 2047];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, k = lane >> 3, j = lane & 7;
    lds_u16 *tab = (lds_u16 *)&s_tab[wave][0];
    const lds_u16 *l_lut = (const lds_u16 *)&s_lut[0];
    const lds_u16 *l_row = (const lds_u16 *)&s_row[0];
    const uint32_t lut0 = (uint32_t)(uintptr_t)l_lut - W3_APM0_LUT_B, tabm2 = (uint32_t)(uintptr_t)tab - 2u;
    const uint32_t onej = 1u << j;
    const uint32_t mult = (uint32_t)__builtin_amdgcn_readfirstlane((int)((1u << a.rate) - 1u));
    const uint32_t rate_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.rate);
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
                const uint32_t p = pc[r];
                // groups (previous byte c1) are contiguous and time ordered; each starts from a fresh table
                const uint64_t vm = __ballot(valid);
                const uint64_t same = __ballot(valid && g == open_g);
                uint32_t o;
                if (same == vm) {
                    const uint64_t lm = vm & 0xFF00FF00FF00FF00ull, em = vm & (vm >> 8) & 0x00FF00FF00FF00FFull;
                    const Apm0Prep q = apm0_prep(tabm2, lut0, p, byte, (uint32_t)j, onej, lm, em);
                    uint32_t au, ao;
                    apm0_commit(q, mult, rate_s, vm, au, ao);
                    o = apm0_finish(q, au, ao);
                } else {
                    // a group boundary inside the round: commit position by position, re-initialising between groups
                    const uint32_t lut = l_lut[p >> 4];
                    const uint32_t c0 = (1u << j) | (byte >> (8 - j));
                    const uint32_t bit = (byte >> (7 - j)) & 1u;
                    const uint32_t upd = c0 * 33u + ((lut & 0x7Eu) >> 1), oth = upd - 1u + ((lut >> 15) << 1);   // the entry trained, and the other one of its pair
                    const uint32_t x = (lut >> 7) & 127u;
                    const int target = bit ? 65535 : 0;
                    int tu = 0, to = 0;
#pragma unroll 1
                    for (int kk = 0; kk < 8; kk++) {
                        if (!((vm >> (kk * 8)) & 1ull)) break;
                        const uint32_t gk = readlane_u32(g, kk * 8);
                        if (gk != open_g) { apm_table_init(tab, l_row, lane); open_g = gk; }
                        if (k == kk) {
                            tu = (int)tab[upd]; to = (int)tab[oth];
                            tab[upd] = (uint16_t)(tu + ((target - tu) >> a.rate));
                        }
                        W3_LDS_FENCE();
                    }
                    const uint32_t pa = (uint32_t)(tu + (((to - tu) * (int)x) >> 7));
                    o = (p + 3u * pa + 2u) >> 2;
                    o = o < 1u ? 1u : o;
                }
                uint16_t *dst = valid ? P + ((uint64_t)rc[r].x * 8u + (uint32_t)j) : a.dummy + lane;
                W3_APM_CHECK_STORE(a, dst, 2u);
                *dst = (uint16_t)o;
            }
        }
    }
}

}  // namespace w3
