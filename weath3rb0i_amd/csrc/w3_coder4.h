// w3_coder4.h — k_coder_x4<L>: the CODE phase with the recurrence wave written as a gfx950 assembly loop.
//
// Same three-wavefront pipeline per 64 blocks as k_coder_x3 (MIX -> RECURRENCE -> OUTPUT through two LDS rings), but
// the serial chain — 8 * block_size dependent bit-steps of arithmetic_coder.rs:41-65 per lane, the block-count-
// independent floor of the whole encoder — is cut to what a lone wavefront can issue fastest.  Measured on MI355X
// (tools/xstep_bench.hip, profiles/r2_xstep_bench.txt): a lone wave pays ~5.4 cycles per instruction SLOT whatever the
// dependency depth, and k_coder_x3 spent 131 cycles per step, 46 of them on per-byte loop code around its 15-instruction
// step (lane masks, address arithmetic, exposed LDS latency).  Here:
//
//   * state is (x1 raw, d = x2 - x1).  One step is 11 VALU instructions:
//       dn  = hi32(d * q + (z:z))           v_mad_u64_u32.  The M-wave pre-bakes the coded bit into the operands:
//                                           bit = 1: q = p32, z = 0           -> dn = floor(d * p32 / 2^32) = m       (:112-116)
//                                           bit = 0: q = 2^32 - p32, z = ~0   -> dn = floor((d * q - 1) / 2^32) = d - m - 1
//                                           (exact: d * p32 mod 2^32 is a multiple of 2^16 and at most 2^32 - 2^16, so adding
//                                           2^64 - 1 to d * q carries into the high dword exactly when it should), i.e. the new
//                                           range without a select
//       x1n = x1 + (dn - d) * z             v_sub, v_mad_u64_u32 (low dword): bit = 0 adds m + 1, bit = 1 adds 0         (:45-48)
//       x2n = x1n + dn
//       s   = clz((x1n ^ x2n) & ((~x1n | x2n) << 1 | 1))      both renormalisation loops in one count (w3_coder.h)
//       x1  = x1n << s ;  d = ((dn + 1) << s) - 1             raw shifts
//     The top-bit fix-ups of loop 2 (x1 &= 0x7FFFFFFF, x2 |= 0x80000000, :59-60) are never executed: they leave d
//     unchanged, bit 31 of x1 falls out of the s formula (it is shifted away by "<< 1"), and the OUTPUT wave rebuilds the
//     true bit 31 of every token from the token before it (raw x1 = previous x1n << previous s).
//   * operands (z, z, q) arrive as ONE ds_read_b96 per step, tokens (x1n, s) leave as one ds_write2st64_b64 per two
//     steps.  A run of full hand-off chunks (W3_X4_CH input bytes each) is ONE asm loop: no lane masks, no address
//     arithmetic per byte, the operands of byte k+1 — also across chunk boundaries — requested before the steps of byte
//     k, the neighbour waves' progress counters fetched two bytes before they are needed.
//   * ragged ends (a lane whose block ends inside a chunk: only the last block of the input) run the same step in C.
//
// The OUTPUT wave checks its accumulator once per input byte (sum of the byte's shift counts) instead of once per step.
// LDS: operands 16 B and tokens 8 B per step and lane -> 12 KiB per ring byte, 144 KiB for the rings of 3 chunks of 4 bytes
// (with two chunks the three waves ran in lockstep and every hand-off latency showed: 21.7 ms instead of 15.8 for the X-wave alone).
#pragma once
#include "w3_coder.h"

namespace w3 {

#define W3_X4_CH 4                    // input bytes per hand-off chunk
#define W3_X4_NCH 3                   // chunks per ring: the M-wave may run two chunks ahead of the X-wave, the X-wave two ahead of the O-wave
#define W3_X4_RING (W3_X4_CH * W3_X4_NCH)   // ring depth in input bytes
#define W3_X4_SYNC_M 0                // byte offsets of the progress words in sync_w
#define W3_X4_SYNC_X 4
#define W3_X4_SYNC_O 8
#define W3_X4_SYNC_ABORT 12

#ifndef W3_X4_EXP
#define W3_X4_EXP 0                   // timing experiments (results WRONG): 1 = O-wave absorbs nothing, 2 = M-wave writes constant operands, 3 = both
#endif

#define W3S_(x) #x
#define W3S(x) W3S_(x)

// one step; operands of step E of the byte in v[B+4E .. B+4E+2] = (z, z, q); token -> TKP = v[TK0:TK1]
// v110 = x1 raw (the pair v[110:111] is the 64-bit addend of the second mad; its high half is never read back)
#define W3_X4_STEP(B, E, TKP, TK0, TK1)                                                                          \
    "v_mad_u64_u32 v[96:97], vcc, %[d], v[" W3S(B) "+4*" W3S(E) "+2], v[" W3S(B) "+4*" W3S(E) ":" W3S(B) "+4*" W3S(E) "+1]\n" \
    "v_sub_u32 v98, v97, %[d]\n"                                                                                 \
    "v_mad_u64_u32 " TKP ", vcc, v98, v[" W3S(B) "+4*" W3S(E) "], v[110:111]\n"                                  \
    "v_add_u32 v98, " TK0 ", v97\n"                                                                              \
    "v_bfi_b32 v99, " TK0 ", v98, -1\n"                                                                          \
    "v_lshl_or_b32 v99, v99, 1, 1\n"                                                                             \
    "v_bitop3_b32 v99, v99, " TK0 ", v98 bitop3:0x60\n"                                                          \
    "v_ffbh_u32 " TK1 ", v99\n"                                                                                  \
    "v_lshlrev_b32 v110, " TK1 ", " TK0 "\n"                                                                     \
    "v_add_u32 v97, 1, v97\n"                                                                                    \
    "v_lshl_add_u32 %[d], v97, " TK1 ", -1\n"
// the 8 operand triples of byte K of the chunk -> v[B .. B+31]   (v106 = this lane's operand address of the chunk)
#define W3_X4_RD(B, K)                                                                       \
    "ds_read_b96 v[" W3S(B) "+0:" W3S(B) "+2], v106 offset:" W3S(K) "*8192+0\n"              \
    "ds_read_b96 v[" W3S(B) "+4:" W3S(B) "+6], v106 offset:" W3S(K) "*8192+1024\n"           \
    "ds_read_b96 v[" W3S(B) "+8:" W3S(B) "+10], v106 offset:" W3S(K) "*8192+2048\n"          \
    "ds_read_b96 v[" W3S(B) "+12:" W3S(B) "+14], v106 offset:" W3S(K) "*8192+3072\n"         \
    "ds_read_b96 v[" W3S(B) "+16:" W3S(B) "+18], v106 offset:" W3S(K) "*8192+4096\n"         \
    "ds_read_b96 v[" W3S(B) "+20:" W3S(B) "+22], v106 offset:" W3S(K) "*8192+5120\n"         \
    "ds_read_b96 v[" W3S(B) "+24:" W3S(B) "+26], v106 offset:" W3S(K) "*8192+6144\n"         \
    "ds_read_b96 v[" W3S(B) "+28:" W3S(B) "+30], v106 offset:" W3S(K) "*8192+7168\n"
// tokens of steps J, J+1 of byte K (v107 = this lane's token address of the chunk; step stride 512 B = one st64 unit)
#define W3_X4_WR(K, J) "ds_write2st64_b64 v107, v[100:101], v[104:105] offset0:" W3S(K) "*8+" W3S(J) " offset1:" W3S(K) "*8+" W3S(J) "+1\n"
#define W3_X4_BYTE(B, K)                                                                                          \
    W3_X4_STEP(B, 0, "v[100:101]", "v100", "v101") W3_X4_STEP(B, 1, "v[104:105]", "v104", "v105") W3_X4_WR(K, 0)    \
    W3_X4_STEP(B, 2, "v[100:101]", "v100", "v101") W3_X4_STEP(B, 3, "v[104:105]", "v104", "v105") W3_X4_WR(K, 2)    \
    W3_X4_STEP(B, 4, "v[100:101]", "v100", "v101") W3_X4_STEP(B, 5, "v[104:105]", "v104", "v105") W3_X4_WR(K, 4)    \
    W3_X4_STEP(B, 6, "v[100:101]", "v100", "v101") W3_X4_STEP(B, 7, "v[104:105]", "v104", "v105") W3_X4_WR(K, 6)
// blocking wait until the progress word at sync_w + OFFS reaches NEED (an SGPR); SEEN (an "s" operand) gets the value read.
// Every spin is bounded and checks the abort word, so the wave always reaches the end of the kernel.
#define W3_X4_SPIN(LBL, OFFS, NEED, SEEN)                          \
    "s_mov_b32 s44, 0\n"                                           \
    LBL "_loop_%=:\n"                                               \
    "ds_read_b32 v108, %[sync] offset:" W3S(OFFS) "\n"             \
    "ds_read_b32 v109, %[sync] offset:" W3S(W3_X4_SYNC_ABORT) "\n" \
    "s_waitcnt lgkmcnt(0)\n"                                       \
    "v_readfirstlane_b32 " SEEN ", v108\n"                         \
    "v_readfirstlane_b32 s45, v109\n"                              \
    "s_cmp_lg_u32 s45, 0\n"                                        \
    "s_cbranch_scc1 Ldead_%=\n"                                     \
    "s_cmp_ge_u32 " SEEN ", " NEED "\n"                            \
    "s_cbranch_scc1 " LBL "_done_%=\n"                              \
    "s_sleep 1\n"                                                  \
    "s_add_u32 s44, s44, 1\n"                                      \
    "s_cmp_lt_u32 s44, 0x1000000\n"                                \
    "s_cbranch_scc1 " LBL "_loop_%=\n"                              \
    "s_branch Ldead_%=\n"                                           \
    LBL "_done_%=:\n"
// operand address of the chunk in ring slot SREG (0 .. NCH-1): v106 = ops_lane + SREG * CH * 8192
#define W3_X4_OPADDR(SREG)                                         \
    "s_lshl_b32 s40, " SREG ", 15\n"                               \
    "v_add_u32 v106, s40, %[opl]\n"
// s47 = ring slot after %[slot]
#define W3_X4_NEXTSLOT                                             \
    "s_add_u32 s47, %[slot], 1\n"                                  \
    "s_cmp_eq_u32 s47, " W3S(W3_X4_NCH) "\n"                       \
    "s_cselect_b32 s47, 0, s47\n"
// DS operations of one wave return in order, so "lgkmcnt(N)" after issuing N of them means: everything older has landed.
// (Measured and rejected: one operand read of the next byte slotted behind the first mad of every step instead of the burst of
// eight at the byte boundary — 16.7 -> 17.8 ms.)
// Codes the chunks [i, iend) (multiples of W3_X4_CH; at least one), publishes x_done for all but the last of them.
#define W3_X4_LOOP                                                                                   \
    "v_mov_b32 v110, %[x1]\n"                                                                        \
    "s_add_u32 s41, %[i], " W3S(W3_X4_CH) "\n"                                                       \
    "s_cmp_ge_u32 %[sm], s41\n"                                                                      \
    "s_cbranch_scc1 Lm0_done_%=\n"                                                                    \
    W3_X4_SPIN("Lm0", W3_X4_SYNC_M, "s41", "%[sm]")                                                  \
    W3_X4_OPADDR("%[slot]")                                                                          \
    W3_X4_RD(32, 0)                                                                                  \
    "Ltop_%=:\n"                                                                                      \
    /* the token slots of this chunk must have been consumed: o_cons + RING >= i + CH */             \
    "s_add_u32 s41, %[i], " W3S(W3_X4_CH) "\n"                                                       \
    "s_add_u32 s42, %[so], " W3S(W3_X4_RING) "\n"                                                    \
    "s_cmp_ge_u32 s42, s41\n"                                                                        \
    "s_cbranch_scc1 Lo_done_%=\n"                                                                     \
    "s_sub_u32 s43, s41, " W3S(W3_X4_RING) "\n"                                                      \
    W3_X4_SPIN("Lo", W3_X4_SYNC_O, "s43", "%[so]")                                                   \
    "s_lshl_b32 s40, %[slot], 14\n"                                                                  \
    "v_add_u32 v107, s40, %[tkl]\n"                                                                  \
    W3_X4_NEXTSLOT                                                                                   \
    W3_X4_RD(64, 1) "s_waitcnt lgkmcnt(8)\n" W3_X4_BYTE(32, 0)                                       \
    W3_X4_RD(32, 2) "s_waitcnt lgkmcnt(8)\n" W3_X4_BYTE(64, 1)                                       \
    W3_X4_RD(64, 3)                                                                                  \
    "ds_read_b32 v108, %[sync] offset:" W3S(W3_X4_SYNC_M) "\n"                                       \
    "ds_read_b32 v109, %[sync] offset:" W3S(W3_X4_SYNC_O) "\n"                                       \
    "s_waitcnt lgkmcnt(10)\n" W3_X4_BYTE(32, 2)                                                      \
    "s_waitcnt lgkmcnt(4)\n"                                                                         \
    "v_readfirstlane_b32 %[sm], v108\n"                                                              \
    "v_readfirstlane_b32 %[so], v109\n"                                                              \
    "s_add_u32 s46, %[i], " W3S(W3_X4_CH) "\n"                                                       \
    "s_cmp_ge_u32 s46, %[iend]\n"                                                                    \
    "s_cbranch_scc1 Llast_%=\n"                                                                       \
    /* another chunk follows: request its first byte's operands before this chunk's last byte is coded */ \
    "s_add_u32 s41, s46, " W3S(W3_X4_CH) "\n"                                                        \
    "s_cmp_ge_u32 %[sm], s41\n"                                                                      \
    "s_cbranch_scc1 Lm1_done_%=\n"                                                                    \
    W3_X4_SPIN("Lm1", W3_X4_SYNC_M, "s41", "%[sm]")                                                  \
    W3_X4_OPADDR("s47")                                                                              \
    W3_X4_RD(32, 0)                                                                                  \
    W3_X4_BYTE(64, 3)                                                                                \
    "v_mov_b32 v108, s46\n"                                                                          \
    "ds_write_b32 %[sync], v108 offset:" W3S(W3_X4_SYNC_X) "\n"                                      \
    "s_mov_b32 %[i], s46\n"                                                                          \
    "s_mov_b32 %[slot], s47\n"                                                                       \
    "s_branch Ltop_%=\n"                                                                              \
    "Llast_%=:\n"                                                                                     \
    W3_X4_BYTE(64, 3)                                                                                \
    "s_mov_b32 %[i], s46\n"                                                                          \
    "s_mov_b32 %[slot], s47\n"                                                                       \
    "s_branch Lexit_%=\n"                                                                             \
    "Ldead_%=:\n"                                                                                     \
    "v_mov_b32 v108, 1\n"                                                                            \
    "ds_write_b32 %[sync], v108 offset:" W3S(W3_X4_SYNC_ABORT) "\n"                                  \
    "s_mov_b32 %[st], 1\n"                                                                           \
    "Lexit_%=:\n"                                                                                     \
    "s_waitcnt lgkmcnt(0)\n"                                                                         \
    "v_mov_b32 %[x1], v110\n"
// ---------------------------------------------------------------------------------------------------------------------
// OUTPUT wave: a run of full chunks as one asm loop.  Per input byte (8 tokens (x_j, s_j) in v[T .. T+15]):
//   S = sum s_j;  if (S > 31 || nb + S > fill) leave to the C path (nothing of the byte has been applied yet)
//   xt_j = x_j ^ (xr & 2^31), xr = x_j << s_j        the TRUE low end of every token (see k_coder_x4's header)
//   U_j  = xt_j >> (31 - s_j)                        carry into the slot | the s_j new bits
//   merge pairwise: (U_a, s_a) . (U_b, s_b) = ((U_a << s_b) + U_b, s_a + s_b);  acc = (acc << S) + U_0..7;  nb += S
//   move 32 finalised bits out when nb >= 33 + (trailing ones of acc)        (never the slot or the pending ones)
// Fixed registers: acc v[84:85], nb v86, pos v87, xr v88, v77 = 0 (high half of the merged value), v96 = token address of the chunk.
#define W3_O4_TOKRD(T, K)                                                                                   \
    "ds_read2st64_b64 v[" W3S(T) "+0:" W3S(T) "+3], v96 offset0:" W3S(K) "*8+0 offset1:" W3S(K) "*8+1\n"      \
    "ds_read2st64_b64 v[" W3S(T) "+4:" W3S(T) "+7], v96 offset0:" W3S(K) "*8+2 offset1:" W3S(K) "*8+3\n"      \
    "ds_read2st64_b64 v[" W3S(T) "+8:" W3S(T) "+11], v96 offset0:" W3S(K) "*8+4 offset1:" W3S(K) "*8+5\n"     \
    "ds_read2st64_b64 v[" W3S(T) "+12:" W3S(T) "+15], v96 offset0:" W3S(K) "*8+6 offset1:" W3S(K) "*8+7\n"
#define W3_O4_XT(T, J, D)                                                                                   \
    "v_bitop3_b32 " D ", v88, v[" W3S(T) "+2*" W3S(J) "], %[k31] bitop3:0x6c\n"                               \
    "v_lshlrev_b32 v88, v[" W3S(T) "+2*" W3S(J) "+1], v[" W3S(T) "+2*" W3S(J) "]\n"
#define W3_O4_U(T, J, D, TMP)                                                                               \
    "v_sub_u32 " TMP ", 31, v[" W3S(T) "+2*" W3S(J) "+1]\n"                                                   \
    "v_lshrrev_b32 " D ", " TMP ", " D "\n"
#define W3_O4_BYTE(T, K)                                                                                    \
    /* widths and the fast-path test first: a byte that leaves for the C path has changed nothing */         \
    "v_add_u32 v73, v[" W3S(T) "+5], v[" W3S(T) "+7]\n"                  /* w23   */                         \
    "v_add_u32 v74, v[" W3S(T) "+13], v[" W3S(T) "+15]\n"                /* w67   */                         \
    "v_add3_u32 v75, v74, v[" W3S(T) "+9], v[" W3S(T) "+11]\n"           /* w4567 */                         \
    "v_add3_u32 v79, v73, v[" W3S(T) "+1], v[" W3S(T) "+3]\n"            /* w0123 */                         \
    "v_add_u32 v79, v79, v75\n"                                          /* S     */                         \
    "v_add_u32 v78, v86, v79\n"                                          /* nb + S */                        \
    "v_cmp_lt_u32 vcc, 31, v79\n"                                                                            \
    "v_cmp_lt_u32 s[52:53], %[fill], v78\n"                                                                  \
    "s_or_b64 vcc, vcc, s[52:53]\n"                                                                          \
    "s_cbranch_vccnz Lbail" W3S(K) "_%=\n"                                                                    \
    W3_O4_XT(T, 0, "v64") W3_O4_XT(T, 1, "v65") W3_O4_XT(T, 2, "v66") W3_O4_XT(T, 3, "v67")                  \
    W3_O4_XT(T, 4, "v68") W3_O4_XT(T, 5, "v69") W3_O4_XT(T, 6, "v70") W3_O4_XT(T, 7, "v71")                  \
    W3_O4_U(T, 0, "v64", "v72") W3_O4_U(T, 1, "v65", "v80") W3_O4_U(T, 2, "v66", "v72") W3_O4_U(T, 3, "v67", "v80") \
    W3_O4_U(T, 4, "v68", "v72") W3_O4_U(T, 5, "v69", "v80") W3_O4_U(T, 6, "v70", "v72") W3_O4_U(T, 7, "v71", "v80") \
    "v_lshl_add_u32 v64, v64, v[" W3S(T) "+3], v65\n"                    /* U01 */                           \
    "v_lshl_add_u32 v66, v66, v[" W3S(T) "+7], v67\n"                    /* U23 */                           \
    "v_lshl_add_u32 v68, v68, v[" W3S(T) "+11], v69\n"                   /* U45 */                           \
    "v_lshl_add_u32 v70, v70, v[" W3S(T) "+15], v71\n"                   /* U67 */                           \
    "v_lshl_add_u32 v64, v64, v73, v66\n"                                /* U03 */                           \
    "v_lshl_add_u32 v68, v68, v74, v70\n"                                /* U47 */                           \
    "v_lshl_add_u32 v76, v64, v75, v68\n"                                /* U07 */                           \
    "v_lshlrev_b64 v[84:85], v79, v[84:85]\n"                                                                \
    "v_lshl_add_u64 v[84:85], v[84:85], 0, v[76:77]\n"                                                       \
    "v_mov_b32 v86, v78\n"                                                                                   \
    "v_not_b32 v80, v84\n"                                                                                   \
    "v_ffbl_b32 v80, v80\n"                                                                                  \
    "v_min_u32 v80, 32, v80\n"                                                                               \
    "v_add_u32 v80, 33, v80\n"                                                                               \
    "v_cmp_ge_u32 vcc, v86, v80\n"                                                                           \
    "s_and_saveexec_b64 s[54:55], vcc\n"                                                                     \
    "s_cbranch_execz Lnf" W3S(K) "_%=\n"                                                                      \
    "v_subrev_u32 v86, 32, v86\n"                                                                            \
    "v_lshrrev_b64 v[80:81], v86, v[84:85]\n"                                                                \
    "v_perm_b32 v80, 0, v80, %[bsw]\n"                                                                       \
    "v_add_u32 v82, 4, v87\n"                                                                                \
    "v_cmp_le_u32 vcc, v82, %[cap]\n"                                                                        \
    "s_and_saveexec_b64 s[56:57], vcc\n"                                                                     \
    "v_add_u32 v83, %[voff], v87\n"                                                                          \
    "global_store_dword v83, v80, %[base]\n"                                                                 \
    "s_mov_b64 exec, s[56:57]\n"                                                                             \
    "v_mov_b32 v87, v82\n"                                                                                   \
    "Lnf" W3S(K) "_%=:\n"                                                                                     \
    "s_or_b64 exec, exec, s[54:55]\n"
// Absorbs the chunks [i, iend); st = 0 done, 1 = pipeline abort, 2 = byte %[k] of chunk %[i] must take the C path.
#define W3_O4_LOOP                                                                                   \
    "v_mov_b32 v84, %[alo]\n v_mov_b32 v85, %[ahi]\n v_mov_b32 v86, %[nb]\n v_mov_b32 v87, %[pos]\n"  \
    "v_mov_b32 v88, %[xr]\n v_mov_b32 v77, 0\n"                                                      \
    "Ltop_%=:\n"                                                                                      \
    "s_add_u32 s41, %[i], " W3S(W3_X4_CH) "\n"                                                       \
    "s_cmp_ge_u32 %[sx], s41\n"                                                                      \
    "s_cbranch_scc1 Lx_done_%=\n"                                                                     \
    W3_X4_SPIN("Lx", W3_X4_SYNC_X, "s41", "%[sx]")                                                   \
    "s_lshl_b32 s40, %[slot], 14\n"                                                                  \
    "v_add_u32 v96, s40, %[tkl]\n"                                                                   \
    W3_X4_NEXTSLOT                                                                                   \
    W3_O4_TOKRD(32, 0) W3_O4_TOKRD(48, 1) "s_waitcnt lgkmcnt(4)\n" W3_O4_BYTE(32, 0)                 \
    W3_O4_TOKRD(32, 2) "s_waitcnt lgkmcnt(4)\n" W3_O4_BYTE(48, 1)                                    \
    W3_O4_TOKRD(48, 3) "s_waitcnt lgkmcnt(4)\n" W3_O4_BYTE(32, 2)                                    \
    "s_waitcnt lgkmcnt(0)\n" W3_O4_BYTE(48, 3)                                                       \
    "v_mov_b32 v108, s41\n"                                                                          \
    "ds_write_b32 %[sync], v108 offset:" W3S(W3_X4_SYNC_O) "\n"                                      \
    "s_mov_b32 %[i], s41\n"                                                                          \
    "s_mov_b32 %[slot], s47\n"                                                                       \
    "s_cmp_lt_u32 %[i], %[iend]\n"                                                                   \
    "s_cbranch_scc1 Ltop_%=\n"                                                                        \
    "s_branch Lexit_%=\n"                                                                             \
    "Lbail0_%=:\n s_mov_b32 %[k], 0\n s_branch Lbail_%=\n"                                             \
    "Lbail1_%=:\n s_mov_b32 %[k], 1\n s_branch Lbail_%=\n"                                             \
    "Lbail2_%=:\n s_mov_b32 %[k], 2\n s_branch Lbail_%=\n"                                             \
    "Lbail3_%=:\n s_mov_b32 %[k], 3\n"                                                                \
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
#define W3_O4_CLOBBERS                                                                                                     \
    "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",         \
    "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",         \
    "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",         \
    "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v96", "v108", "v109",                                   \
    "s40", "s41", "s44", "s45", "s47", "s52", "s53", "s54", "s55", "s56", "s57", "vcc", "scc", "memory"
#define W3_X4_CLOBBERS                                                                                                     \
    "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",         \
    "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",         \
    "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",         \
    "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",         \
    "v96", "v97", "v98", "v99", "v100", "v101", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",             \
    "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "vcc", "scc", "memory"

// ring position (in bytes) of input byte j: chunks take the ring's NCH slots in turn
__device__ __forceinline__ uint32_t x4_ring_pos(uint32_t j) { return (j / W3_X4_CH) % W3_X4_NCH * W3_X4_CH + j % W3_X4_CH; }

struct X4Op { uint32_t z0, z1, q, pad; };   // 16-byte operand slot; the X-wave reads the first 12 bytes

// The step in C: ragged chunks of the X-wave (same operands, same tokens as the asm form).
__device__ __forceinline__ uint2 x4_step_c(uint32_t &x1, uint32_t &d, const uint32_t z, const uint32_t q) {
    const uint64_t prod = (uint64_t)d * q + (((uint64_t)z << 32) | z);
    const uint32_t dn = (uint32_t)(prod >> 32);
    const uint32_t x1n = x1 + (dn - d) * z;
    const uint32_t x2n = x1n + dn;
    const uint32_t s = (uint32_t)__builtin_clz((x1n ^ x2n) & (((~x1n | x2n) << 1) | 1u));
    x1 = x1n << s;
    d = ((dn + 1u) << s) - 1u;
    return make_uint2(x1n, s);
}

typedef uint32_t w3_u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t w3_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint8_t w3_lds_u8;

// L > 1 leaf streams: the mix makes the M-wave's chunk longer than the X-wave's, so TWO M-waves take alternate chunks
// (workgroup = M0, M1, X, O: one wave per SIMD of the CU); with a single stream one M-wave keeps up (51 vs 72 cycles per step).
template <int L>
__global__ void __launch_bounds__(L > 1 ? 256 : 192) k_coder_x4(Coder3Args a) {
    constexpr uint32_t NM = L > 1 ? 2u : 1u;   // M-waves
    __shared__ X4Op opq[W3_X4_RING * 8u * 64u];    // M -> X: (z, z, q) per step           [ring byte][bit][lane]
    __shared__ uint2 tok[W3_X4_RING * 8u * 64u];   // X -> O: (x1n raw, s) per step        [ring byte][bit][lane]
    __shared__ uint2 fin[64];                      // X -> O: (x1 raw, d) after the lane's last step
    __shared__ uint32_t sync_w[8];                 // [0] M produced, [1] X done, [2] O consumed, [3] abort   (bytes)
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
    constexpr uint32_t CH = W3_X4_CH, RING = W3_X4_RING;

    if (wave < NM) {
        // ------------------------------ M-wave(s) ------------------------------
        // Operands are fetched several chunks ahead with unconditional loads (hipcc waits vmcnt(0) after a load it has to branch
        // around): a lane streams its own block, 64 B of probabilities per chunk, so every second chunk opens a new line
        // with a full memory latency, and a chunk is coded in under a microsecond.  Positions past the end of a lane's block
        // are clamped to its last byte: the X-wave codes them like any others and the O-wave ignores the tokens.
        const uint32_t last = (act && len) ? len - 1u : 0u;
        const uint32_t ops_lane = (uint32_t)(uintptr_t)(w3_lds_u8 *)(opq + lane);
        // the chunk's CH input bytes as ONE unaligned dword load at min(i, len - CH) (never past the block's end); shifted into place after
        const uint64_t blk_end4 = off + (uint64_t)len >= CH ? off + (uint64_t)len - CH : 0ull;   // n >= 4: stays inside the input
        static_assert(CH == 4, "the byte fetch below is one dword");
        struct Buf { uint4 p[L][CH]; uint32_t bytes, sh; };
        auto load = [&](Buf &bf, uint32_t i0) {
#if W3_X4_EXP & 4
            bf.bytes = i0 * 2654435761u; bf.sh = 0;
            for (uint32_t k = 0; k < CH; k++) for (int l = 0; l < L; l++) bf.p[l][k] = make_uint4(i0 + 0x12345u, i0 * 77u + 0x4567u, i0 + 0x333u, i0 + 0x9999u);
            return;
#endif
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
#if !(W3_X4_EXP & 16)
            if (i >= RING && seen + RING < i + CH) {    // this chunk's slots still hold bytes [i - RING, i - RING + CH): they must have been coded
                seen = spin_until_ge<1>(x_done, i + CH - RING, abortf, dead);
                if (dead) return;
            }
#endif
            const uint32_t nbytes4 = ~(bf.sh < 32u ? bf.bytes >> bf.sh : 0u);   // complemented: z = ~0 when the coded bit is 0; byte k at bits 8k..8k+7
            const uint32_t slot = ops_lane + (x4_ring_pos(i) << 13);
#if W3_X4_EXP & 2
            if (i < RING) {
                w3_u32x3 o0; o0.x = 0u; o0.y = 0u; o0.z = 0x80000000u;
#pragma unroll
                for (uint32_t k = 0; k < CH * 8; k++) asm volatile("ds_write_b96 %0, %1 offset:%2" : : "v"(slot), "v"(o0), "n"(k * 1024) : "memory");
            }
#else
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
                    w3_u32x3 o0, o1;
                    o0.x = z0; o0.y = z0; o0.z = ((w0 << 16) ^ z0) - z0;            // bit ? p32 : 2^32 - p32
                    o1.x = z1; o1.y = z1; o1.z = ((w0 & 0xFFFF0000u) ^ z1) - z1;
#if W3_X4_EXP & 8
                    asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(slot), "v"(o0.z ^ o0.x), "n"((k * 8 + 2 * q) * 1024) : "memory");
                    asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(slot), "v"(o1.z ^ o1.x), "n"((k * 8 + 2 * q + 1) * 1024) : "memory");
#else
                    asm volatile("ds_write_b96 %0, %1 offset:%2" : : "v"(slot), "v"(o0), "n"((k * 8 + 2 * q) * 1024) : "memory");
                    asm volatile("ds_write_b96 %0, %1 offset:%2" : : "v"(slot), "v"(o1), "n"((k * 8 + 2 * q + 1) * 1024) : "memory");
#endif
                }
            }
#endif
            __asm__ volatile("" ::: "memory");
            if (NM > 1u && i > 0u) {   // chunks are published in order: the other M-wave's chunk before this one
                (void)spin_until_ge<1>(m_prod, i, abortf, dead);
                if (dead) return;
            }
            lds_store_u32(m_prod, min(i + CH, maxlen));   // after the operands: the LDS executes one wave's operations in order
        };
        // four buffers in flight per wave: a chunk's loads are issued three of this wave's chunk-times (>= 3 us) before its
        // operands are built.  M-wave w takes the chunks w, w + NM, w + 2 NM, ...
        constexpr uint32_t ST = NM * CH;   // bytes between two chunks of one M-wave
        const uint32_t i0 = wave * CH;
        Buf bA, bB, bC, bD;
        load(bA, i0); load(bB, i0 + ST); load(bC, i0 + 2u * ST); load(bD, i0 + 3u * ST);
        for (uint32_t i = i0; i < maxlen && !dead; i += 4u * ST) {
            produce(bA, i);
            load(bA, i + 4u * ST);
            if (dead || i + ST >= maxlen) break;
            produce(bB, i + ST);
            load(bB, i + 5u * ST);
            if (dead || i + 2u * ST >= maxlen) break;
            produce(bC, i + 2u * ST);
            load(bC, i + 6u * ST);
            if (dead || i + 3u * ST >= maxlen) break;
            produce(bD, i + 3u * ST);
            load(bD, i + 7u * ST);
        }
        return;
    }

    if (wave == NM) {
        // ------------------------------ X-wave ------------------------------
#if W3_X4_EXP & 64
        return;
#endif
        __builtin_amdgcn_s_setprio(3);
        const uint32_t ops_lane = (uint32_t)(uintptr_t)(w3_lds_u8 *)(opq + lane);
        const uint32_t tok_lane = (uint32_t)(uintptr_t)(w3_lds_u8 *)(tok + lane);
        const uint32_t sync_addr = (uint32_t)(uintptr_t)(w3_lds_u8 *)sync_w;
        uint32_t x1 = 0u, d = 0xFFFFFFFFu;
        uint32_t seen_m = 0, seen_o = 0, i = 0;
        const uint32_t full_end = maxlen / CH * CH;
        const bool raggedB = lenB % CH != 0u;
        while (i < maxlen && !dead) {
            // [i, run_end): full chunks in which no lane's block ends except at run_end itself
            uint32_t run_end = full_end;
            if (lenB > i) run_end = min(run_end, lenB / CH * CH);   // the shorter lanes end at (or inside the chunk after) this boundary
            if (run_end > i) {
                uint32_t status = 0, slot = (i / CH) % W3_X4_NCH;
                asm volatile(W3_X4_LOOP
                             : [d] "+v"(d), [x1] "+v"(x1), [i] "+s"(i), [slot] "+s"(slot), [sm] "+s"(seen_m), [so] "+s"(seen_o), [st] "+s"(status)
                             : [iend] "s"(run_end), [opl] "v"(ops_lane), [tkl] "v"(tok_lane), [sync] "v"(sync_addr)
                             : W3_X4_CLOBBERS);
                if (status) { dead = true; break; }
                if (len == run_end) fin[lane] = make_uint2(x1, d);
                __asm__ volatile("" ::: "memory");
                lds_store_u32(x_done, run_end);   // (the asm loop leaves the last chunk's hand-off to us: fin goes first)
            } else {
                // a chunk in which some lane's block ends before the chunk does (or the block's last, partial chunk)
                const uint32_t need = min(i + CH, maxlen);
                if (seen_m < need) { seen_m = __builtin_amdgcn_readfirstlane(spin_until_ge(m_prod, need, abortf, dead)); if (dead) break; }
                if (seen_o + RING < i + CH) { seen_o = __builtin_amdgcn_readfirstlane(spin_until_ge(o_cons, i + CH - RING, abortf, dead)); if (dead) break; }
                for (uint32_t k = 0; k < CH && i + k < maxlen; k++) {
                    const size_t ring = ((size_t)x4_ring_pos(i + k) * 8u) * 64u + lane;
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
            (void)raggedB;
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
    const uint32_t full_end = maxlen / CH * CH;

    // one byte's eight tokens the careful way: per-step accumulator guard, hand-back to k_coder when a pending run outgrows it
    auto byte_c = [&](uint32_t j0) {
        const uint2 *slot = tok + ((size_t)x4_ring_pos(j0) * 8u) * 64u + lane;
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
        uint32_t run_end = full_end;
        if (lenB > i) run_end = min(run_end, lenB / CH * CH);
#if W3_X4_EXP & 1
        {
            const uint32_t need = min(i + CH, maxlen);
#if !(W3_X4_EXP & 32)
            if (seen_x < need) { seen_x = __builtin_amdgcn_readfirstlane(spin_until_ge<1>(x_done, need, abortf, dead)); if (dead) break; }
#endif
            if (need == maxlen && act) a.out_len[b] = 0u;
            lds_store_u32(o_cons, need);
            i += CH;
            continue;
        }
#endif
        if (run_end > i) {
            uint32_t status = 0, slot = (i / CH) % W3_X4_NCH, kbail = 0;
            uint32_t alo = (uint32_t)acc, ahi = (uint32_t)(acc >> 32);
            asm volatile(W3_O4_LOOP
                         : [alo] "+v"(alo), [ahi] "+v"(ahi), [nb] "+v"(nb), [pos] "+v"(pos), [xr] "+v"(xr), [i] "+s"(i), [slot] "+s"(slot),
                           [sx] "+s"(seen_x), [st] "+s"(status), [k] "+s"(kbail)
                         : [iend] "s"(run_end), [tkl] "v"(tok_lane), [sync] "v"(sync_addr), [cap] "v"(cap), [voff] "v"(voff),
                           [base] "s"(wg_base), [fill] "s"(fast_fill), [k31] "s"(0x80000000u), [bsw] "s"(0x00010203u)
                         : W3_O4_CLOBBERS);
            acc = ((uint64_t)ahi << 32) | alo;
            if (status == 1u) { dead = true; break; }
            if (status == 2u) {   // bytes kbail.. of chunk i the careful way (x_done already covers the chunk)
                for (uint32_t k = kbail; k < CH; k++) byte_c(i + k);
                if (len == i + CH) finish();
                __asm__ volatile("" ::: "memory");
                lds_store_u32(o_cons, i + CH);
                i += CH;
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
