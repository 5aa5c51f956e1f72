// w3_coder5.h — k_coder_x5<L>: k_coder_x4's three-wavefront CODE pipeline (MIX -> RECURRENCE -> OUTPUT, the recurrence and
// output waves as gfx950 assembly loops; see w3_coder4.h for the step itself) with HALF the LDS: rings of three 2-byte
// chunks instead of three 4-byte chunks, 72.5 KiB per workgroup instead of 144.5 KiB.
//
// Why: the coder is one latency chain per lane (8 * block_size dependent steps, arithmetic_coder.rs:41-65) on 239 of the
// 256 CUs, three wavefronts each.  It cannot be made shorter by more hardware, but the hardware it leaves idle can work on
// the NEXT call's predict phase (w3_encode_submit / w3_encode_wait) — if those wavefronts find LDS beside it.  With 144 KiB
// per workgroup 16 KiB per CU were left; with 72.5 KiB a predict workgroup of 82,000 bytes fits (w3_predict.h,
// "half-CU" kernels).
//
// What changes against k_coder_x4:
//   * a ring is ONE revolution of the asm loops: 6 input bytes = 3 chunks, every LDS offset a constant of the instruction
//     (no slot arithmetic); a run of revolutions is one asm loop.  The last < 6 bytes of a block (65,536 = 6 * 10,922 + 4)
//     and ragged lanes run the same step in C, as before.
//   * progress words are exchanged per 2-byte chunk; the X-wave refreshes its view of the neighbours' words once per
//     chunk (two ds_reads issued ahead of a byte's steps, consumed after it).
//   * the M-wave builds a chunk's 16 operands in registers BEFORE it waits for the chunk's ring slot (a slot is free for
//     1.5 chunk-times only: the wait must not be followed by the mix), and keeps 8 / 6 / 4 two-byte buffers of loads in
//     flight (L = 1 / 2 / >= 3).  L > 1: two M-waves take alternate chunks and publish in order.
// Tokens, operands, the O-wave's accumulator, the hand-back to the robust coder and the counting sink are k_coder_x4's.
#pragma once
#include "w3_coder4.h"

namespace w3 {

#define W3_X5_CH 2                    // input bytes per hand-off chunk
#ifndef W3_X5_NB1
#define W3_X5_NB1 8u                  // chunk buffers the M-wave of a one-stream coder keeps in flight
#endif
#define W3_X5_RING 6                  // ring depth in input bytes = one revolution of the asm loops (3 chunks)

// operands of ring byte K -> v[B .. B+30]; %[opl] = this lane's operand address (constant: the ring never moves)
#define W3_X5_RD(B, K)                                                                          \
    "ds_read_b96 v[" W3S(B) "+0:" W3S(B) "+2], %[opl] offset:" W3S(K) "*8192+0\n"              \
    "ds_read_b96 v[" W3S(B) "+4:" W3S(B) "+6], %[opl] offset:" W3S(K) "*8192+1024\n"           \
    "ds_read_b96 v[" W3S(B) "+8:" W3S(B) "+10], %[opl] offset:" W3S(K) "*8192+2048\n"          \
    "ds_read_b96 v[" W3S(B) "+12:" W3S(B) "+14], %[opl] offset:" W3S(K) "*8192+3072\n"         \
    "ds_read_b96 v[" W3S(B) "+16:" W3S(B) "+18], %[opl] offset:" W3S(K) "*8192+4096\n"         \
    "ds_read_b96 v[" W3S(B) "+20:" W3S(B) "+22], %[opl] offset:" W3S(K) "*8192+5120\n"         \
    "ds_read_b96 v[" W3S(B) "+24:" W3S(B) "+26], %[opl] offset:" W3S(K) "*8192+6144\n"         \
    "ds_read_b96 v[" W3S(B) "+28:" W3S(B) "+30], %[opl] offset:" W3S(K) "*8192+7168\n"
#define W3_X5_WR(K, J) "ds_write2st64_b64 %[tkl], v[100:101], v[104:105] offset0:" W3S(K) "*8+" W3S(J) " offset1:" W3S(K) "*8+" W3S(J) "+1\n"
#define W3_X5_BYTE(B, K)                                                                                          \
    W3_X4_STEP(B, 0, "v[100:101]", "v100", "v101") W3_X4_STEP(B, 1, "v[104:105]", "v104", "v105") W3_X5_WR(K, 0)    \
    W3_X4_STEP(B, 2, "v[100:101]", "v100", "v101") W3_X4_STEP(B, 3, "v[104:105]", "v104", "v105") W3_X5_WR(K, 2)    \
    W3_X4_STEP(B, 4, "v[100:101]", "v100", "v101") W3_X4_STEP(B, 5, "v[104:105]", "v104", "v105") W3_X5_WR(K, 4)    \
    W3_X4_STEP(B, 6, "v[100:101]", "v100", "v101") W3_X4_STEP(B, 7, "v[104:105]", "v104", "v105") W3_X5_WR(K, 6)
// the M-wave must have produced byte i + AHEAD (exclusive): cached view first, then the blocking poll
#define W3_X5_NEED_M(LBL, AHEAD)                                   \
    "s_add_u32 s41, %[i], " W3S(AHEAD) "\n"                        \
    "s_cmp_ge_u32 %[sm], s41\n"                                    \
    "s_cbranch_scc1 " LBL "_done_%=\n"                              \
    W3_X4_SPIN(LBL, W3_X4_SYNC_M, "s41", "%[sm]")
// the O-wave must have consumed the tokens this chunk (ending at i + END) overwrites: o_cons + RING >= i + END
#define W3_X5_NEED_O(LBL, END)                                     \
    "s_add_u32 s42, %[so], " W3S(W3_X5_RING) "-(" W3S(END) ")\n"   \
    "s_cmp_ge_u32 s42, %[i]\n"                                     \
    "s_cbranch_scc1 " LBL "_done_%=\n"                              \
    "s_sub_u32 s43, %[i], " W3S(W3_X5_RING) "-(" W3S(END) ")\n"    \
    W3_X4_SPIN(LBL, W3_X4_SYNC_O, "s43", "%[so]")
#define W3_X5_REFRESH_RD                                                          \
    "ds_read_b32 v102, %[sync] offset:" W3S(W3_X4_SYNC_M) "\n"                   \
    "ds_read_b32 v103, %[sync] offset:" W3S(W3_X4_SYNC_O) "\n"
#define W3_X5_REFRESH_USE                                                         \
    "v_readfirstlane_b32 %[sm], v102\n"                                          \
    "v_readfirstlane_b32 %[so], v103\n"
#define W3_X5_PUBLISH(END)                                                        \
    "s_add_u32 s46, %[i], " W3S(END) "\n"                                        \
    "v_mov_b32 v108, s46\n"                                                      \
    "ds_write_b32 %[sync], v108 offset:" W3S(W3_X4_SYNC_X) "\n"
// One chunk of the revolution: bytes K0 (operands already requested into buffer A = v32..) and K0+1 (buffer B = v64..);
// NEXT = the ring byte whose operands are requested for the chunk after (AHEAD = its end, relative to i).
// DS operations of one wave return in order: "lgkmcnt(N)" after issuing N of them means everything older has landed.
#define W3_X5_CHUNK_HEAD(Q, K0, K1)                                                                  \
    W3_X5_NEED_O("Lo" W3S(Q), K1 + 1)                                                                \
    W3_X5_RD(64, K1)                                                                                 \
    W3_X5_REFRESH_RD                                                                                 \
    "s_waitcnt lgkmcnt(10)\n" W3_X5_BYTE(32, K0)                                                     \
    "s_waitcnt lgkmcnt(4)\n"                                                                         \
    W3_X5_REFRESH_USE
#define W3_X5_CHUNK_TAIL(Q, K1, NEXT, AHEAD)                                                         \
    W3_X5_NEED_M("Lm" W3S(Q), AHEAD)                                                                 \
    W3_X5_RD(32, NEXT)                                                                               \
    W3_X5_BYTE(64, K1)                                                                               \
    W3_X5_PUBLISH(K1 + 1)
// Codes the revolutions [i, iend) (iend - i a positive multiple of 6, i a multiple of 6); publishes x_done for every chunk but the last.
#define W3_X5_LOOP                                                                                   \
    "v_mov_b32 v110, %[x1]\n"                                                                        \
    W3_X5_NEED_M("Lmp", 2)                                                                           \
    W3_X5_RD(32, 0)                                                                                  \
    "Ltop_%=:\n"                                                                                      \
    W3_X5_CHUNK_HEAD(0, 0, 1) W3_X5_CHUNK_TAIL(0, 1, 2, 4)                                           \
    W3_X5_CHUNK_HEAD(1, 2, 3) W3_X5_CHUNK_TAIL(1, 3, 4, 6)                                           \
    W3_X5_CHUNK_HEAD(2, 4, 5)                                                                        \
    "s_add_u32 s47, %[i], " W3S(W3_X5_RING) "\n"                                                     \
    "s_cmp_ge_u32 s47, %[iend]\n"                                                                    \
    "s_cbranch_scc1 Llast_%=\n"                                                                       \
    W3_X5_CHUNK_TAIL(2, 5, 0, 8)                                                                     \
    "s_mov_b32 %[i], s47\n"                                                                          \
    "s_branch Ltop_%=\n"                                                                              \
    "Llast_%=:\n"                                                                                     \
    W3_X5_BYTE(64, 5)                                                                                \
    "s_mov_b32 %[i], s47\n"                                                                          \
    "s_branch Lexit_%=\n"                                                                             \
    "Ldead_%=:\n"                                                                                     \
    "v_mov_b32 v108, 1\n"                                                                            \
    "ds_write_b32 %[sync], v108 offset:" W3S(W3_X4_SYNC_ABORT) "\n"                                  \
    "s_mov_b32 %[st], 1\n"                                                                           \
    "Lexit_%=:\n"                                                                                     \
    "s_waitcnt lgkmcnt(0)\n"                                                                         \
    "v_mov_b32 %[x1], v110\n"
#define W3_X5_CLOBBERS W3_X4_CLOBBERS, "v102", "v103"

// OUTPUT wave, one revolution per loop trip: token address %[tkl] constant, eight tokens of ring byte K in v[T .. T+15]
#define W3_O5_TOKRD(T, K)                                                                                       \
    "ds_read2st64_b64 v[" W3S(T) "+0:" W3S(T) "+3], %[tkl] offset0:" W3S(K) "*8+0 offset1:" W3S(K) "*8+1\n"      \
    "ds_read2st64_b64 v[" W3S(T) "+4:" W3S(T) "+7], %[tkl] offset0:" W3S(K) "*8+2 offset1:" W3S(K) "*8+3\n"      \
    "ds_read2st64_b64 v[" W3S(T) "+8:" W3S(T) "+11], %[tkl] offset0:" W3S(K) "*8+4 offset1:" W3S(K) "*8+5\n"     \
    "ds_read2st64_b64 v[" W3S(T) "+12:" W3S(T) "+15], %[tkl] offset0:" W3S(K) "*8+6 offset1:" W3S(K) "*8+7\n"
#define W3_O5_CHUNK(Q, K0, K1)                                                                       \
    "s_add_u32 s41, %[i], " W3S(K1) "+1\n"                                                           \
    "s_cmp_ge_u32 %[sx], s41\n"                                                                      \
    "s_cbranch_scc1 Lx" W3S(Q) "_done_%=\n"                                                           \
    W3_X4_SPIN("Lx" W3S(Q), W3_X4_SYNC_X, "s41", "%[sx]")                                            \
    W3_O5_TOKRD(32, K0) W3_O5_TOKRD(48, K1) "s_waitcnt lgkmcnt(4)\n" W3_O4_BYTE(32, K0)              \
    "s_waitcnt lgkmcnt(0)\n" W3_O4_BYTE(48, K1)                                                      \
    "v_mov_b32 v108, s41\n"                                                                          \
    "ds_write_b32 %[sync], v108 offset:" W3S(W3_X4_SYNC_O) "\n"
// Absorbs the revolutions [i, iend); st = 0 done, 1 = pipeline abort, 2 = ring byte %[k] of the revolution at %[i] must take the C path
// (the chunks before it have been absorbed and published).
#define W3_O5_LOOP                                                                                   \
    "v_mov_b32 v84, %[alo]\n v_mov_b32 v85, %[ahi]\n v_mov_b32 v86, %[nb]\n v_mov_b32 v87, %[pos]\n"  \
    "v_mov_b32 v88, %[xr]\n v_mov_b32 v77, 0\n"                                                      \
    "Ltop_%=:\n"                                                                                      \
    W3_O5_CHUNK(0, 0, 1) W3_O5_CHUNK(1, 2, 3) W3_O5_CHUNK(2, 4, 5)                                   \
    "s_add_u32 %[i], %[i], " W3S(W3_X5_RING) "\n"                                                    \
    "s_cmp_lt_u32 %[i], %[iend]\n"                                                                   \
    "s_cbranch_scc1 Ltop_%=\n"                                                                        \
    "s_branch Lexit_%=\n"                                                                             \
    "Lbail0_%=:\n s_mov_b32 %[k], 0\n s_branch Lbail_%=\n"                                             \
    "Lbail1_%=:\n s_mov_b32 %[k], 1\n s_branch Lbail_%=\n"                                             \
    "Lbail2_%=:\n s_mov_b32 %[k], 2\n s_branch Lbail_%=\n"                                             \
    "Lbail3_%=:\n s_mov_b32 %[k], 3\n s_branch Lbail_%=\n"                                             \
    "Lbail4_%=:\n s_mov_b32 %[k], 4\n s_branch Lbail_%=\n"                                             \
    "Lbail5_%=:\n s_mov_b32 %[k], 5\n"                                                                \
    "Lbail_%=:\n"                                                                                     \
    "s_mov_b32 %[st], 2\n"                                                                           \
    "s_branch Lexit_%=\n"                                                                             \
    "Ldead_%=:\n"                                                                                     \
    "v_mov_b32 v108, 1\n"                                                                            \
    "ds_write_b32 %[sync], v108 offset:" W3S(W3_X4_SYNC_ABORT) "\n"                                  \
    "s_mov_b32 %[st], 1\n"                                                                           \
    "Lexit_%=:\n"                                                                                     \
    "s_waitcnt lgkmcnt(0)\n"                                                                         \
    "v_mov_b32 %[alo], v84\n v_mov_b32 %[ahi], v85\n v_mov_b32 %[nb], v86\n v_mov_b32 %[pos], v87\n v_mov_b32 %[xr], v88\n"

template <int L>
__global__ void __launch_bounds__(L > 1 ? 256 : 192) k_coder_x5(Coder3Args a) {
    constexpr uint32_t NM = L > 1 ? 2u : 1u;   // M-waves
    constexpr uint32_t CH = W3_X5_CH, RING = W3_X5_RING;
    __shared__ X4Op opq[RING * 8u * 64u];    // M -> X: (z, z, q) per step           [ring byte][bit][lane]
    __shared__ uint2 tok[RING * 8u * 64u];   // X -> O: (x1n raw, s) per step        [ring byte][bit][lane]
    __shared__ uint2 fin[64];                // X -> O: (x1 raw, d) after the lane's last step
    __shared__ uint32_t sync_w[8];           // [0] M produced, [1] X done, [2] O consumed, [3] abort   (bytes)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t b = blockIdx.x * 64u + lane;
    const bool act = b < a.nblocks;
    const uint64_t off = (uint64_t)(act ? b : 0u) * a.block_size;
    const uint32_t len = act ? (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size) : 0u;
    uint32_t maxlen = len, lenB = len ? len : 0xFFFFFFFFu;
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) {
        maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, dd, 64));
        lenB = min(lenB, (uint32_t)__shfl_xor((int)lenB, dd, 64));
    }
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    lenB = __builtin_amdgcn_readfirstlane(lenB);     // the shorter of the (at most two) block lengths in this wave; 0 < lenB <= maxlen
    if (threadIdx.x < 8) sync_w[threadIdx.x] = 0u;
    __syncthreads();
    volatile uint32_t *m_prod = &sync_w[0], *x_done = &sync_w[1], *o_cons = &sync_w[2], *abortf = &sync_w[3];
    bool dead = false;
    // [i, asm_end(i)): whole revolutions in which no lane's block ends except at the end of the last one; i must start a revolution
    auto asm_end = [&](uint32_t i) -> uint32_t {
        uint32_t run_end = maxlen;
        if (lenB > i) run_end = min(run_end, lenB);   // the shorter lanes end at (or inside the revolution after) this boundary
        return i % RING == 0u ? i + (run_end - i) / RING * RING : i;
    };

    if (wave != NM && a.prio_mo) __builtin_amdgcn_s_setprio(2);
    if (wave < NM) {
        // ------------------------------ M-wave(s) ------------------------------
        // Loads are issued NB chunks of this wave ahead (unconditional, clamped: hipcc waits vmcnt(0) after a load it has to branch
        // around).  Positions past the end of a lane's block are clamped to its last byte: the X-wave codes them like any
        // others and the O-wave ignores the tokens.
        const uint32_t last = (act && len) ? len - 1u : 0u;
        const uint32_t ops_lane = (uint32_t)(uintptr_t)(w3_lds_u8 *)(opq + lane);
        const uint64_t blk_end4 = off + (uint64_t)len >= 4u ? off + (uint64_t)len - 4u : 0ull;   // n >= 4: stays inside the input
        struct Buf { uint4 p[L][CH]; uint32_t bytes, sh; };
        auto load = [&](Buf &bf, uint32_t i0) {
            // the chunk's input bytes as ONE unaligned dword load at min(i0, len - 4) (never past the block's end); shifted into place after
            const uint64_t want = off + i0, at = want < blk_end4 ? want : blk_end4;
            uint32_t w; __builtin_memcpy(&w, a.in + at, 4);
            bf.bytes = w; bf.sh = (uint32_t)(want - at) * 8u;
#pragma unroll
            for (uint32_t k = 0; k < CH; k++) {
                const uint32_t ic = min(i0 + k, last);
#pragma unroll
                for (int l = 0; l < L; l++) bf.p[l][k] = a.src[l][off + ic];
            }
        };
        uint32_t seen = 0;   // last value read from x_done
        auto produce = [&](const Buf &bf, uint32_t i) {
            // 1. the chunk's 16 operands, in registers
            const uint32_t nbytes4 = ~(bf.sh < 32u ? bf.bytes >> bf.sh : 0u);   // complemented: z = ~0 when the coded bit is 0; byte k at bits 8k..8k+7
            uint32_t qv[CH * 8], zv[CH * 8];
#pragma unroll
            for (uint32_t k = 0; k < CH; k++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {   // one dword = two steps
                    uint32_t w0 = q == 0 ? bf.p[0][k].x : q == 1 ? bf.p[0][k].y : q == 2 ? bf.p[0][k].z : bf.p[0][k].w;
                    if constexpr (L > 1) {   // OpinionMixer2, both steps of the dword at once (see k_coder_x3)
                        u16x2 P = as_u16x2(w0), D = pk_opinion_dist(P);
#pragma unroll
                        for (int l = 1; l < L; l++) {
                            const uint32_t w = q == 0 ? bf.p[l][k].x : q == 1 ? bf.p[l][k].y : q == 2 ? bf.p[l][k].z : bf.p[l][k].w;
                            const u16x2 Q = as_u16x2(w), E = pk_opinion_dist(Q);
                            const uint32_t mask = pk_farther_mask(D, E);                   // 0xFFFF where E > D
                            P = as_u16x2((as_u32(Q) & mask) | (as_u32(P) & ~mask));
                            D = __builtin_elementwise_max(D, E);
                        }
                        w0 = as_u32(P);
                    }
                    const uint32_t z0 = (uint32_t)__builtin_amdgcn_sbfe((int)nbytes4, 8 * k + 7 - 2 * q, 1);
                    const uint32_t z1 = (uint32_t)__builtin_amdgcn_sbfe((int)nbytes4, 8 * k + 6 - 2 * q, 1);
                    zv[k * 8 + 2 * q] = z0; qv[k * 8 + 2 * q] = ((w0 << 16) ^ z0) - z0;                 // bit ? p32 : 2^32 - p32
                    zv[k * 8 + 2 * q + 1] = z1; qv[k * 8 + 2 * q + 1] = ((w0 & 0xFFFF0000u) ^ z1) - z1;
                }
            }
            // 2. the chunk's ring slots still hold bytes [i - RING, i - RING + CH): they must have been coded
            if (i >= RING && seen + RING < i + CH) {
                seen = spin_until_ge<1>(x_done, i + CH - RING, abortf, dead);
                if (dead) return;
            }
            // 3. sixteen ds_write_b96
            const uint32_t slot = ops_lane + ((i % RING) << 13);
#pragma unroll
            for (uint32_t e = 0; e < CH * 8; e++) {
                w3_u32x3 o; o.x = zv[e]; o.y = zv[e]; o.z = qv[e];
                asm volatile("ds_write_b96 %0, %1 offset:%2" : : "v"(slot), "v"(o), "n"(e * 1024) : "memory");
            }
            __asm__ volatile("" ::: "memory");
            if (NM > 1u && i > 0u) {   // chunks are published in order: the other M-wave's chunk before this one
                (void)spin_until_ge<1>(m_prod, i, abortf, dead);
                if (dead) return;
            }
            lds_store_u32(m_prod, min(i + CH, maxlen));   // after the operands: the LDS executes one wave's operations in order
        };
        constexpr uint32_t NB = L == 1 ? W3_X5_NB1 : L == 2 ? 6u : 4u;   // chunk buffers in flight per wave
        constexpr uint32_t ST = NM * CH;                          // bytes between two chunks of one M-wave
        const uint32_t i0 = wave * CH;
        Buf bf[NB];
#pragma unroll
        for (uint32_t k = 0; k < NB; k++) load(bf[k], i0 + k * ST);
        for (uint32_t i = i0; i < maxlen && !dead; i += NB * ST) {
#pragma unroll
            for (uint32_t k = 0; k < NB; k++) {
                if (dead || i + k * ST >= maxlen) break;
                produce(bf[k], i + k * ST);
                load(bf[k], i + (NB + k) * ST);
            }
        }
        return;
    }

    if (wave == NM) {
        // ------------------------------ X-wave ------------------------------
        __builtin_amdgcn_s_setprio(3);
        const uint32_t ops_lane = (uint32_t)(uintptr_t)(w3_lds_u8 *)(opq + lane);
        const uint32_t tok_lane = (uint32_t)(uintptr_t)(w3_lds_u8 *)(tok + lane);
        const uint32_t sync_addr = (uint32_t)(uintptr_t)(w3_lds_u8 *)sync_w;
        uint32_t x1 = 0u, d = 0xFFFFFFFFu;
        uint32_t seen_m = 0, seen_o = 0, i = 0;
        while (i < maxlen && !dead) {
            const uint32_t run_end = asm_end(i);
            if (run_end > i) {
                uint32_t status = 0;
                asm volatile(W3_X5_LOOP
                             : [d] "+v"(d), [x1] "+v"(x1), [i] "+s"(i), [sm] "+s"(seen_m), [so] "+s"(seen_o), [st] "+s"(status)
                             : [iend] "s"(run_end), [opl] "v"(ops_lane), [tkl] "v"(tok_lane), [sync] "v"(sync_addr)
                             : W3_X5_CLOBBERS);
                if (status) { dead = true; break; }
                if (len == run_end) fin[lane] = make_uint2(x1, d);
                __asm__ volatile("" ::: "memory");
                lds_store_u32(x_done, run_end);   // (the asm loop leaves the last chunk's hand-off to us: fin goes first)
            } else {
                // a chunk outside a whole revolution: the block's tail, or one in which some lane's block ends
                const uint32_t need = min(i + CH, maxlen);
                if (seen_m < need) { seen_m = __builtin_amdgcn_readfirstlane(spin_until_ge(m_prod, need, abortf, dead)); if (dead) break; }
                if (seen_o + RING < i + CH) { seen_o = __builtin_amdgcn_readfirstlane(spin_until_ge(o_cons, i + CH - RING, abortf, dead)); if (dead) break; }
                for (uint32_t k = 0; k < CH && i + k < maxlen; k++) {
                    const size_t ring = ((size_t)((i + k) % RING) * 8u) * 64u + lane;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const X4Op op = opq[ring + j * 64];
                        tok[ring + j * 64] = x4_step_c(x1, d, op.z0, op.q);
                    }
                    if (i + k + 1u == len) fin[lane] = make_uint2(x1, d);
                }
                __asm__ volatile("" ::: "memory");
                lds_store_u32(x_done, need);
                i += CH;
            }
        }
        return;
    }

    // -------------------------------- O-wave --------------------------------
    uint8_t *out = a.stripes + (uint64_t)(act ? b : 0u) * a.stripe_cap;
    uint32_t cap = act ? a.stripe_cap : 0u;
    const uint32_t limit = a.acc_limit, fast_fill = a.acc_limit + 18u;   // 64 for the default limit of 46
    uint64_t acc = 0ull; uint32_t nb = 1u, pos = 0u;
    uint32_t xr = 0u;   // the raw x1 the next token grew from (its bit 31 is all that matters)
    uint32_t failed = 0u;
    uint32_t seen_x = 0, i = 0;
    const uint32_t tok_lane = (uint32_t)(uintptr_t)(w3_lds_u8 *)(tok + lane);
    const uint32_t sync_addr = (uint32_t)(uintptr_t)(w3_lds_u8 *)sync_w;
    const uint8_t *wg_base = a.stripes + (uint64_t)blockIdx.x * 64u * a.stripe_cap;   // stripe of this workgroup's first block
    const uint32_t voff = lane * a.stripe_cap;                                       // (64 stripes: below 2^32 for every block size)

    // one byte's eight tokens the careful way: per-step accumulator guard, hand-back to k_coder when a pending run outgrows it
    auto byte_c = [&](uint32_t j0) {
        const uint2 *slot = tok + ((size_t)(j0 % RING) * 8u) * 64u + lane;
#pragma unroll 1
        for (int j = 0; j < 8; j++) {
            const uint2 t = slot[j * 64];
            if (nb > limit) {
                // accumulator nearly full: drain finalised bytes (those above the slot) one at a time
                const uint32_t pend = trailing_ones64(acc) + 1u;
#pragma unroll 1
                while (nb >= pend + 8u) {
                    const uint8_t v = (uint8_t)(acc >> (nb - 8u));
                    if (pos < cap) out[pos] = v;
                    pos += 1u; nb -= 8u;
                }
                if (nb > limit) { failed = 1u; acc = 0ull; nb = 1u; }   // pending run longer than the accumulator: k_coder re-codes the block
            }
            const uint32_t xt = t.x ^ (xr & 0x80000000u), sj = t.y;   // the TRUE low end
            xr = t.x << sj;
            acc += xt >> 31;
            acc = (acc << sj) | __builtin_amdgcn_ubfe(xt, 31u - sj, sj);
            nb += sj;
        }
        // once per input byte: move 32 finalised bits out (never the slot or the pending ones)
        const uint32_t lo = (uint32_t)acc;
        const uint32_t pend = (~lo ? (uint32_t)__builtin_ctz(~lo) : 32u) + 1u;
        if (nb >= pend + 32u) {
            const uint32_t wv = (uint32_t)(acc >> (nb - 32u));
            if (pos + 4u <= cap) { const uint32_t be = __builtin_bswap32(wv); __builtin_memcpy(out + pos, &be, 4); }
            pos += 4u; nb -= 32u;
        }
    };
    // ArithmeticCoder::flush -> ACWriter::flush(x2) (arithmetic_coder.rs:67-71, io.rs:91-100): first bit x2 >> 31 (= 1) resolves
    // the slot and the pending bits, then x2's next bits pad to a byte
    auto finish = [&]() {
        const uint2 f = fin[lane];
        const uint32_t x2f = ((f.x & 0x7FFFFFFFu) + f.y) | 0x80000000u;
        if (a.out_bits && !failed && cap) a.out_bits[b] = 8u * pos + nb - (trailing_ones64(acc) + 1u);   // ACStats (helpers.rs:60-90): all bits but the slot and the pending ones
        uint64_t fa = acc + 1ull; uint32_t fnb = nb, fpos = pos;
        const uint32_t idx = fnb & 7u;
        if (idx) { const uint32_t kk = 8u - idx; fa = (fa << kk) | ((x2f << 1) >> (32u - kk)); fnb += kk; }
#pragma unroll 1
        while (fnb >= 8u) {
            const uint8_t v = (uint8_t)(fa >> (fnb - 8u));
            if (fpos < cap) out[fpos] = v;
            fpos += 1u; fnb -= 8u;
        }
        if (failed) { const uint32_t kk = atomicAdd(&a.flags[1], 1u); a.redo[kk] = b; }
        else { a.out_len[b] = fpos; if (fpos > cap) atomicOr(&a.flags[0], 1u); }
        cap = 0u;   // the lane keeps absorbing the tokens of clamped operands; nothing of it is stored any more
    };

    while (i < maxlen && !dead) {
        const uint32_t run_end = asm_end(i);
        if (run_end > i) {
            uint32_t status = 0, kbail = 0;
            uint32_t alo = (uint32_t)acc, ahi = (uint32_t)(acc >> 32);
            asm volatile(W3_O5_LOOP
                         : [alo] "+v"(alo), [ahi] "+v"(ahi), [nb] "+v"(nb), [pos] "+v"(pos), [xr] "+v"(xr), [i] "+s"(i),
                           [sx] "+s"(seen_x), [st] "+s"(status), [k] "+s"(kbail)
                         : [iend] "s"(run_end), [tkl] "v"(tok_lane), [sync] "v"(sync_addr), [cap] "v"(cap), [voff] "v"(voff),
                           [base] "s"(wg_base), [fill] "s"(fast_fill), [k31] "s"(0x80000000u), [bsw] "s"(0x00010203u)
                         : W3_O4_CLOBBERS);
            acc = ((uint64_t)ahi << 32) | alo;
            if (status == 1u) { dead = true; break; }
            if (status == 2u) {   // ring byte kbail of the revolution at i the careful way, to the end of its chunk (x_done already covers it)
                const uint32_t j0 = i + kbail, cend = (j0 / CH + 1u) * CH;
                for (uint32_t j = j0; j < cend; j++) byte_c(j);
                if (len == cend) finish();
                __asm__ volatile("" ::: "memory");
                lds_store_u32(o_cons, cend);
                i = cend;
                continue;
            }
            if (len == run_end) finish();
        } else {
            const uint32_t need = min(i + CH, maxlen);
            if (seen_x < need) { seen_x = __builtin_amdgcn_readfirstlane(spin_until_ge<1>(x_done, need, abortf, dead)); if (dead) break; }
            __asm__ volatile("" ::: "memory");
            for (uint32_t k = 0; i + k < need; k++) {
                byte_c(i + k);
                if (i + k + 1u == len) finish();
            }
            __asm__ volatile("" ::: "memory");
            lds_store_u32(o_cons, need);
            i += CH;
        }
    }
    if (dead && lane == 0) atomicOr(&a.flags[0], 2u);
}

}  // namespace w3
