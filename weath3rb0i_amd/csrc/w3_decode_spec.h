// w3_decode_spec.h — the DECODER of Counter-leaf models (+ APM chain) with the nibble's whole context tree evaluated at once.
//
// Decoding is serial: the context of bit t+1 contains bit t (main.rs:131-140).  k_generic_nl / k_cm_nl run that loop literally, one lane
// per block: per BIT a round trip for the leaves' Counters and a dependent one for every APM stage, 239 wavefronts on 256 CUs at enwik9
// size — 7.6 us per bit-step, 250 MiB/s, latency from end to end.  But the contexts a NIBBLE can reach are known when it starts: bit k of
// the nibble has 2^k candidate contexts (the k bits decoded before it), 15 in all, and — with alignment_bits >= 2 (t mod 4 is part of every
// context) and APM rows keyed by the partial byte — no two of them share a Counter or an APM entry.  So a block is decoded by SIXTEEN lanes:
//   lane r < 15 = node (k, x) of the nibble's tree (k = floor(log2(r + 1)), x = r + 1 - 2^k): it forms the node's context of every leaf
//   (OrderN::update / OrderNEntropy::update in closed form, leaf_ctx of w3_generic.h, on the history  hist << k | x), looks the Counters up
//   (models/counter.rs), mixes (OpinionMixer2) and runs the APM chain — all 15 nodes side by side, ONE round trip per table level per
//   nibble instead of four;
//   then the four bits are decoded one after the other (arithmetic_coder.rs:74-106; the decoder state is replicated in the 16 lanes), each
//   step taking the probability of the node the bits so far select (ds_bpermute inside the 16-lane row);
//   the four nodes on the decoded path update their Counters (Counter::update) and APM entries; an exact-map slot is claimed by
//   compare-and-swap (two path nodes may meet one empty slot).
// A wavefront holds four blocks: 3,815 wavefronts at enwik9 size instead of 239, and the chip hides the round trips it still has.
// Model tables: the lane-per-block decoder's layout (layout_generic / layout_cm), one region per block.
// Covered: up to 8 leaves — Counter-table leaves of any history kind with alignment_bits >= 2 (FrozenModel leaves too), slot-state leaves (HS
// below) — and 0..2 APM stages.  Everything else (alignment 0 / 1: the nibble's own updates feed its later contexts; longer chains) stays on
// the lane-per-block kernels, which also remain the cross-check (W3_OPT_VARIANT bit 1024).
#pragma once
#include "w3_cm.h"

namespace w3 {

__device__ __forceinline__ uint32_t ds_ld32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ds_ld64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ds_ld16(const uint16_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ds_st32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ds_st16(uint16_t *p, uint16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Direct-indexed Counter tables and APM tables are only ever touched by plain loads and stores of ONE wavefront (no compare-and-swap at
// the L2): they may live in the CU's L1, which that wavefront's own stores go through.  (The exact maps keep the L2-level accesses above:
// their keys are claimed with atomics, which the L1 does not see.)
__device__ __forceinline__ uint32_t pl_ld32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void pl_st32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void pl_st16(uint16_t *p, uint16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Bucketed exact map, SIXTEEN LANES PER BLOCK (D = 4): one load per lane.  Lane r of the row fetches slot r of the family's bucket — one
// 128-byte request per row instead of sixteen 8-byte loads per lane (with those the kernel was no longer bound by HBM traffic, which the
// buckets had cut from 7.2 to 2.9 TB per 1e9 B, but by its load instructions: profiles/r4_decode/) — and DECODES the key it finds: a stored
// context (window << 3 | bit position) names its own node of the nibble's tree (bit position & 3 = k, the window's low k bits = x) and the
// history it belongs to (the window's other bits); if that is this nibble's history the holder posts {slot + 1, counts} to the node's lane
// through a 512-byte LDS board.  A lane is through when its context was posted (found), or when the bucket has an empty slot (absent: a path
// node claims from there); a full bucket without it makes the whole row go on to the next bucket.
__device__ __forceinline__ void bucket_lookup16(uint32_t *tbl, uint32_t bmask, uint32_t fam, uint64_t hn, uint32_t H, uint32_t half, uint32_t mykey,
                                                uint32_t lane, uint32_t r, uint32_t row0, uint64_t *board, uint32_t *&slot, uint32_t &val, uint32_t &ins) {
    uint32_t hb = (fam * 2654435761u) ^ (fam >> 13);
    bool done = false;
    ins = 0xFFFFFFFFu;
    (void)mykey;
    for (;;) {
        hb &= bmask;
        const uint64_t kv = ds_ld64(reinterpret_cast<const uint64_t *>(tbl + 32u * hb) + r);
        const uint32_t key = (uint32_t)kv;
        __asm__ volatile("" ::: "memory");
        board[lane] = 0ull;
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        {
            const uint32_t c = key - 1u, bp = c & 7u, kn = bp & 3u, W = c >> 3;
            const uint32_t xh = W & ((1u << kn) - 1u), Rh = W >> kn;
            const bool cand = key != 0u && (bp >> 2) == half && Rh == ((uint32_t)hn & ((1u << (H - kn)) - 1u));
            if (cand) board[row0 + (1u << kn) - 1u + xh] = ((uint64_t)(hb * 16u + r + 1u) << 32) | (kv >> 32);
        }
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const uint64_t e = board[lane];
        const uint32_t empties = (uint32_t)(__ballot(key == 0u) >> row0) & 0xFFFFu;
        if (!done) {
            if ((uint32_t)(e >> 32)) { slot = tbl + 2u * ((uint32_t)(e >> 32) - 1u) + 1u; val = (uint32_t)e; done = true; }
            else if (empties) { ins = hb * 16u + (uint32_t)(__ffs((int)empties) - 1); done = true; }
        }
        const uint32_t pending = (uint32_t)(__ballot(!done) >> row0) & 0xFFFFu;   // some lane of my row has to look into the next bucket: the row goes on together
        if (!__ballot(pending != 0u)) break;
        if (pending) hb++;
    }
}

// The Counter look-up of one leaf of k_decode_spec's node (its own macro: the kernel issues these before it stages the slot leaves' Cells).
// Direct tables are NIBBLE-MAJOR (this kernel's own layout; the table is scratch, zeroed per call): a raw-history context of alignment 3 is a
// window of H = bits - 3 history bits plus the bit position.  For node (k, x) of a nibble that starts with history h the window is
// [v : 3-k bits][g : H-3 bits][x : k bits], g = the low H-3 bits of h, v the 3-k bits above them — so the 32 Counters {(k, v, x)} of one
// (half, g) are everything ANY nibble starting with those H-3 history bits can touch: ONE 128-byte line instead of 15 scattered words (the PMC
// passes of the first version: 4.3 KB fetched per nibble and block, HBM-bound at 5 TB/s).  A group may start inside the nibble (D = 2: tb bits
// of it are decoded already).  Exact maps {ctx + 1, counts}: found, or absent (the empty slot ends the probe; only a path node claims one).
#define W3_DS_COUNTER_LOOKUP \
                    if constexpr (RAW) ctx[l] = t_n == 0u ? 0u : ((((uint32_t)hist_n & lp[l].hist_mask) << 3) | (t_n & 7u)); \
                    else ctx[l] = leaf_ctx(lp[l], hist_n, t_n, hs, g.huff); \
                    uint32_t *tbl = reinterpret_cast<uint32_t *>(blk_tbl + lp[l].tbl_off); \
                    if (!lp[l].use_hash) { \
                        uint32_t idx = ctx[l]; \
                        if (RAW || (lp[l].hist <= W3_HIST_RAW && lp[l].align == 3u && lp[l].bits >= 6u)) { \
                            const uint32_t H = lp[l].bits - 3u; \
                            const uint32_t tb = t & 3u, kn = tb + k, xn = (((uint32_t)hist64 & ((1u << tb) - 1u)) << k) | x; \
                            const uint64_t hn = hist64 >> tb; \
                            const uint32_t gq = (uint32_t)hn & ((1u << (H - 3u)) - 1u), v = (uint32_t)(hn >> (H - 3u)) & ((8u >> kn) - 1u); \
                            idx = (((((t >> 2) & 1u) << (H - 3u)) | gq) << 5) | (kn << 3) | (v << kn) | xn; \
                        } \
                        slot[l] = tbl + idx; val[l] = pl_ld32(slot[l]); \
                    } \
                    else if (NM && (RAW || (lp[l].hist <= W3_HIST_RAW && lp[l].align == 3u && lp[l].bits >= 6u))) { \
                        /* NIBBLE-MAJOR BUCKETS of the exact map (raw history, alignment 3): the 15 candidate contexts of a nibble differ only in   \
                           the bits the nibble decodes, so they share (half, g) — hash THAT to a 128-byte bucket of 16 slots {ctx + 1, counts}: one   \
                           line per nibble and leaf instead of 15 random 8-byte probes (1.5 KB of lines, profiles/r3_decode_spec/README.md).        \
                           Found, or absent (a bucket with an empty slot ends the probe); a full bucket without the key continues in the next one. */ \
                        const uint32_t H = lp[l].bits - 3u; \
                        const uint32_t tb = t & 3u; \
                        const uint64_t hn = hist64 >> tb; \
                        const uint32_t fam = ((((t >> 2) & 1u) << (H - 3u)) | ((uint32_t)hn & ((1u << (H - 3u)) - 1u))) + 1u; \
                        const uint32_t bmask = lp[l].hash_mask >> 4; \
                        if constexpr (D == 4) { \
                            bucket_lookup16(tbl, bmask, fam, hn, H, (t >> 2) & 1u, ctx[l] + 1u, lane, r, row0, s_board, slot[l], val[l], ins[l]); \
                        } else { \
                        uint32_t hb = (fam * 2654435761u) ^ (fam >> 13); \
                        ins[l] = 0xFFFFFFFFu; \
                        for (;;) { \
                            hb &= bmask; \
                            const uint64_t *bk = reinterpret_cast<const uint64_t *>(tbl + 32u * hb); \
                            uint64_t kv[16]; \
                            _Pragma("unroll") for (int sl_ = 0; sl_ < 16; sl_++) kv[sl_] = ds_ld64(bk + sl_); \
                            bool hit = false; uint32_t first_empty = 16u; \
                            _Pragma("unroll") for (int sl_ = 15; sl_ >= 0; sl_--) { \
                                const uint32_t key = (uint32_t)kv[sl_]; \
                                if (key == ctx[l] + 1u) { slot[l] = tbl + 32u * hb + 2u * (uint32_t)sl_ + 1u; val[l] = (uint32_t)(kv[sl_] >> 32); hit = true; } \
                                if (key == 0u) first_empty = (uint32_t)sl_; \
                            } \
                            if (hit) break; \
                            if (first_empty < 16u) { ins[l] = hb * 16u + first_empty; break; }   /* absent: a path node claims from here on */ \
                            hb++; \
                        } \
                        } \
                    } \
                    else { \
                        uint32_t h = (ctx[l] * 2654435761u) ^ (ctx[l] >> 15); \
                        for (;;) { \
                            h &= lp[l].hash_mask; \
                            const uint64_t kv = ds_ld64(reinterpret_cast<const uint64_t *>(tbl + 2u * h)); \
                            const uint32_t key = (uint32_t)kv; \
                            if (key == ctx[l] + 1u) { slot[l] = tbl + 2u * h + 1u; val[l] = (uint32_t)(kv >> 32); break; } \
                            if (key == 0u) { if constexpr (NM) ins[l] = h; break; }   /* absent: a path node claims from here on */ \
                            h++; \
                        } \
                    }

// D = bits per speculated group: 4 (the nibble, 16 lanes per block: the shipped form) or 2 (half a nibble, 4 lanes per block, 16 blocks per
// wavefront: 6 instead of 15 speculative look-ups per nibble and leaf, but four round trips per nibble instead of two — measured, no gain:
// see decode_group_bits in w3hip.hip; a tested variant, W3_OPT_TUNE bit 14).
// HS = the spec has slot-state leaves (build-defined CM leaf over hashmap.rs / state_table: DESIGN.md 2.4): such a leaf is nibble-major by nature
// — one Cell per nibble (hashslots.md:3-4), the 15 states of the nibble's tree in one slot — so lane 0 of the row does the Cell's tag match /
// eviction (slot_select of w3_cm.h on the byte-exact 96-byte Cell, hashmap.rs:42-71) when the nibble starts, every node lane reads its 12-bit
// state (hashmap.rs:86-97) and looks its probability up in the state table (staged in LDS), and lane 0 walks the four path states on
// (hashmap.rs:99-112: the packed states share bytes, so one lane writes them one after the other).  The Cell is staged in LDS for the
// nibble (six 16-byte loads by six lanes, w3_cm.h's cmc_* helpers on the copy, six stores back).  D = 4 only.
// NM = the nibble-major table formats of LARGE batches (bucketed exact maps, APM tables [group][j][node]): they cut the HBM traffic a large
// batch is bound by (7.2 -> 3.0 TB per 1e9 B) at the price of a longer dependent chain per nibble (the look-up's LDS exchange), which is
// what a SMALL batch — a few wavefronts per SIMD, nothing to hide latency behind — is bound by: below W3_DECODE_NM_MIN_BLOCKS the round-3
// formats (slot-by-slot probes, row-major APM tables) stay.  A template parameter, not a run-time flag: the flag's branches alone cost the
// small batches 18 % (profiles/r4_decode/).
template <int NL, int NA, int D, bool HS = false, bool NM = false, bool RAW = false, int KM = 0>
__global__ void __launch_bounds__(64) k_decode_spec(CmArgs a) {
    // RAW: every Counter leaf is live, over raw history, alignment 3, bits >= 6, and bit l of KM says whether leaf l is a slot-state leaf — all known
    // at compile time (decode_spec_all_raw / launch_decode_spec)
    static_assert(!RAW || (D == 4 && NL <= 8 && HS == (KM != 0)), "RAW: all-raw-history models, nibble groups");
    static_assert(RAW || KM == 0, "the leaf-kind mask belongs to the RAW instances");
    constexpr uint32_t LPB = 1u << D, BPW = 64u / LPB;        // lanes per block, blocks per wavefront
    static_assert(!HS || D == 4, "slot-state leaves: one Cell per nibble");
    __shared__ int16_t s_str[NA > 0 ? 4096 : 1];
    if (NA > 0) for (uint32_t i = threadIdx.x; i < 4096u; i += 64u) s_str[i] = a.stretch[i];
    __shared__ uint2 s_st[HS ? kStSize : 1];
    if (HS) for (uint32_t i = threadIdx.x; i < (uint32_t)kStSize; i += 64u) s_st[i] = a.st[i];
    // the nibble's Cell of every (row, slot leaf), staged in w3_cm.h's layout: "lane" v = 8 * row + leaf of [chunk][64] x 16 B
    __shared__ cm_u32x4 s_cell[HS ? 6 : 1][HS ? 64 : 1];
    __shared__ uint32_t s_sid[HS ? 64 : 1];   // the selected slot of every staged Cell
    __shared__ LeafParam s_leaf[W3_MAX_LEAVES];
    __shared__ ApmParam s_apm[W3_MAX_APM];
    __shared__ uint64_t s_board[64];   // bucket_lookup16: what the bucket's holders post to the node lanes
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < W3_MAX_APM; k++) s_apm[k] = a.apm[k];
    }
    stage_leaves(s_leaf, a.g);   // (also the barrier for the tables above)
    const GenericArgs &g = a.g;
    // up to four leaves: their parameters straight from the kernel arguments (scalar loads, hoisted out of the loops); the eight-leaf instances keep them in LDS
    const LeafParam *lp = (RAW || (!HS && NL <= 4)) ? a.g.leaf : s_leaf;   // (the general eight-leaf instances from the kernel arguments: measured, no difference)
    auto is_slot = [&](int l) -> bool { return RAW ? ((KM >> l) & 1) != 0 : lp[l].kind == 1; };
    auto is_ctr = [&](int l) -> bool { return RAW ? ((KM >> l) & 1) == 0 : (lp[l].kind == 0 && !lp[l].frozen); };
    const uint32_t lane = threadIdx.x & 63u, r = lane & (LPB - 1u), grp = lane / LPB;
    const uint32_t bl = blockIdx.x * BPW + grp;              // block of this row of LPB lanes inside the batch
    const bool live = bl < g.n_lanes;
    const uint32_t blc = live ? bl : g.n_lanes - 1u;
    const uint32_t b = g.first_block + blc;
    const uint64_t off = (uint64_t)b * g.block_size;
    uint32_t len = (uint32_t)((g.n - off) < g.block_size ? (g.n - off) : g.block_size);
    if (!live) len = 0u;
    uint8_t *blk_tbl = g.tables + (uint64_t)blc * g.lane_stride;
    // this lane's node of the group's tree
    const uint32_t rr = r < LPB - 1u ? r : LPB - 2u;         // (the last lane of a row shadows its neighbour and never updates anything)
    const uint32_t k = 31u - (uint32_t)__builtin_clz(rr + 1u), x = rr + 1u - (1u << k);
    const bool node = r < LPB - 1u;
    const uint32_t row0 = lane & ~(LPB - 1u);
    bool fence_each = false;   // (D = 2: consecutive groups differ in t mod 4, so even alignment 2 never reads what the group before stored)
#pragma unroll
    for (int l = 0; l < NL; l++) fence_each |= !RAW && D == 4 && is_ctr(l) && lp[l].align < 3;
    if (HS) fence_each = true;   // (two nibbles in a row may hash to one Cell: what the first stored must have landed)
    const bool leader = HS && r < (uint32_t)NL && (RAW ? ((KM >> r) & 1u) != 0u : lp[r < (uint32_t)NL ? r : 0u].kind == 1);
    Decoder dec;
    dec.init(g.cin + g.coffs[b], live ? g.clens[b] : 0u);

    uint64_t hist64 = 0; uint32_t t = 0, c0 = 1u, c1 = 0u;
    HuffState hs;
    for (uint32_t i = 0; i < len; i++) {
        for (int gi = 0; gi < 8 / D; gi++) {
            // ---- the node's prediction: contexts, Counters, mix, APM chain ----
            const uint64_t hist_n = (hist64 << k) | x;
            const uint32_t t_n = t + k;
            uint32_t *slot[NL]; uint32_t val[NL], ctx[NL];
            uint32_t ins[NL];   // bucketed exact maps: where an absent context's claim starts (bucket * 16 + first empty slot), or ~0
            uint32_t p = 32768u, best = 0u;
            uint8_t *cellp[HS ? NL : 1]; uint32_t sid[HS ? NL : 1];
            cm_u32x4 cq[HS ? NL : 1];
            if (HS) {
                // the nibble's Cell of every slot-state leaf: lanes 0..5 of the row fetch its six 16-byte chunks into LDS, lane 0 selects the slot
                // there (and evicts on a miss: cmc_select of w3_cm.h, hashmap.rs:42-71), the row learns the slot's id
                const uint32_t tb8 = t & 7u;
                const uint64_t hb = hist64 >> tb8;   // the completed bytes
#pragma unroll
                for (int l = 0; l < NL; l++) {
                    cellp[l] = nullptr; sid[l] = 0u;
                    if (is_slot(l)) {
                        const uint64_t h = slot_hash(lp[l].order, hb, tb8 != 0u, (uint32_t)hist64 & 15u);
                        cellp[l] = blk_tbl + lp[l].tbl_off + (h >> (64u - lp[l].log_cells)) * 96ull;
                        if (r < 6u) cq[l] = *reinterpret_cast<const cm_u32x4 *>(cellp[l] + 16u * r);   // (in flight beside the Counter look-ups below)
                    }
                }
            }
            // ---- Counter-table leaves: contexts, look-ups ----
#pragma unroll
            for (int l = 0; l < NL; l++) {
                slot[l] = nullptr; val[l] = 0u; ctx[l] = 0u; ins[l] = 0xFFFFFFFFu;
                if (is_ctr(l)) {
                    W3_DS_COUNTER_LOOKUP
                }
            }
            if (HS) {
                const uint32_t tb8 = t & 7u;
                const uint64_t hb = hist64 >> tb8;
#pragma unroll
                for (int l = 0; l < NL; l++)
                    if (is_slot(l) && r < 6u) s_cell[r][grp * 8u + (uint32_t)l] = cq[l];
                __asm__ volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                // lane r of the row is the LEADER of leaf r: the slot leaves' selects run side by side (one code path, several lanes), not one
                // after the other on lane 0
                if (leader) {
                    uint32_t my_order = 0u;
                    if constexpr (RAW) {   // (the parameters sit in scalar registers: a select over the leaves, not an index)
#pragma unroll
                        for (int l = 0; l < NL; l++) if (is_slot(l) && r == (uint32_t)l) my_order = lp[l].order;
                    } else my_order = lp[r].order;
                    const uint64_t h = slot_hash(my_order, hb, tb8 != 0u, (uint32_t)hist64 & 15u);
                    s_sid[grp * 8u + r] = cmc_select((cm_lds_u32 *)&s_cell[0][grp * 8u + r], (uint32_t)h & 0xFFFu, s_st);
                }
                __asm__ volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int l = 0; l < NL; l++)
                    if (is_slot(l)) sid[l] = s_sid[grp * 8u + (uint32_t)l];
            }
            if (HS) {
#pragma unroll
                for (int l = 0; l < NL; l++)
                    if (is_slot(l)) {
                        CmCellRef cr; uint32_t cv;   // this node's 12-bit state from the staged Cell (Slot::get_nib path, hashmap.rs:114-121)
                        val[l] = cmc_state((cm_lds_u32 *)&s_cell[0][grp * 8u + (uint32_t)l], slot_idx(sid[l], k, x), cr, cv);
                    }
            }
            bool first_leaf = true;
#pragma unroll
            for (int l = 0; l < NL; l++) {
                if (!RAW && lp[l].kind > 1) continue;   // (NL is an upper bound: unused entries)
                const uint32_t pl = (HS && is_slot(l)) ? (s_st[val[l]].x & 0xFFFFu)      // StateTable::p, state_table/mod.rs:47-49
                                  : RAW ? counter_p_packed(val[l])
                                  : lp[l].frozen ? 32768u : counter_p_packed(val[l]);
                const uint32_t d = opinion_dist(pl);
                if (first_leaf || d > best) { p = pl; best = d; first_leaf = false; }
            }
            const uint32_t c0_n = (c0 << k) | x;
            uint16_t *aslot[NA > 0 ? NA : 1]; int tv[NA > 0 ? NA : 1];
#pragma unroll
            for (int s = 0; s < NA; s++) {   // APM chain (build-defined), as k_cm_nl
                // NIBBLE-MAJOR APM table (this kernel's own layout of the stage's 256 rows per previous-byte page: [17 groups][33][16 node columns]):
                // group 0 = the first nibble's rows 1 .. 15, group 1 + h = the second nibble's 15 rows after high nibble h; the column is the node
                // of the nibble's tree.  The 15 entries a nibble's candidates read for one j are 30 contiguous bytes, so the nibble touches the lines
                // of the j values its candidates' probabilities fall on (confident predictions cluster at the ends) instead of 15 rows 66 bytes apart.
                uint32_t G, col;
                if (c0_n < 16u) { G = 0u; col = c0_n - 1u; }
                else {
                    const uint32_t kk2 = 27u - (uint32_t)__builtin_clz(c0_n);        // bits decoded inside the second nibble
                    G = 1u + ((c0_n >> kk2) & 15u); col = (1u << kk2) - 1u + (c0_n & ((1u << kk2) - 1u));
                }
                const uint32_t page = s_apm[s].ctx_kind ? c1 : 0u;
                const uint32_t pos = (uint32_t)((int)s_str[p >> 4] + 2048) * 32u;
                const uint32_t j = pos >> 12, w = pos & 4095u;
                constexpr bool rowmajor = !NM;
                constexpr uint32_t estride = rowmajor ? 1u : 16u;   // distance (in entries) between t[j] and t[j + 1]
                uint16_t *tr = reinterpret_cast<uint16_t *>(blk_tbl + s_apm[s].off) +
                               (rowmajor ? ((s_apm[s].ctx_kind ? (c0_n | (c1 << 8)) : c0_n) * 33u + j) : ((((page * 17u + G) * 33u + j) << 4) + col));
                const uint32_t v0 = tr[0], v1 = tr[estride];
                const uint32_t pa = (v0 * (4096u - w) + v1 * w) >> 12;
                aslot[s] = tr + ((w >> 11) ? estride : 0u);
                tv[s] = (int)((w >> 11) ? v1 : v0);
                const uint32_t o = (p + 3u * pa + 2u) >> 2;
                p = o < 1u ? 1u : o > 65535u ? 65535u : o;
            }
            // ---- the nibble's four bits, one after the other ----
            uint32_t xk = 0u, mybit = 0u;
            bool on = false;
#pragma unroll
            for (uint32_t kk = 0; kk < (uint32_t)D; kk++) {
                const uint32_t src = row0 | ((1u << kk) - 1u + xk);
                const uint32_t psel = (uint32_t)__shfl((int)p, (int)src, 64);
                const uint32_t bit = (RAW && !HS) ? dec.decode_nz(psel) : dec.decode(psel);   // (Counter::p and the APM clamp give 1 .. 65535)
                if (node && k == kk && x == xk) { on = true; mybit = bit; }
                xk = (xk << 1) | bit;
            }
            // ---- the path's nodes adapt (Counter::update, counter.rs:20-26; the APM entry nearer to the looked-up position) ----
            if (HS) {
                // slot-state leaves: each leaf's leader lane walks the four path states on in the staged Cell (the packed states share bytes: one
                // lane per Cell, one state after the other), then lanes 0..5 write the Cell back
                __asm__ volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (leader) {   // (every slot leaf's leader at once)
                    cm_lds_u32 *cbv = (cm_lds_u32 *)&s_cell[0][grp * 8u + r];
                    const uint32_t my_sid = s_sid[grp * 8u + r];
#pragma unroll
                    for (uint32_t kk = 0; kk < 4u; kk++) {
                        const uint32_t pre = xk >> (4u - kk);                                   // the nibble's first kk bits
                        const uint32_t bitk = (xk >> (3u - kk)) & 1u;
                        CmCellRef cr; uint32_t cv;
                        const uint32_t idx = slot_idx(my_sid, kk, pre);
                        const uint32_t stv = cmc_state(cbv, idx, cr, cv);
                        cmc_set_state(cbv, idx, cr, cv, bitk ? (s_st[stv].y & 0xFFFFu) : (s_st[stv].x >> 16));
                        __asm__ volatile("" ::: "memory");
                    }
                }
                __asm__ volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int l = 0; l < NL; l++)
                    if (is_slot(l) && r < 6u) *reinterpret_cast<cm_u32x4 *>(cellp[l] + 16u * r) = s_cell[r][grp * 8u + (uint32_t)l];
            }
            uint32_t claim_rank[NL];
#pragma unroll
            for (int l = 0; l < NL; l++) {   // (wave-uniform conditions: every lane takes part in the ballots)
                claim_rank[l] = 0u;
                if (NM && is_ctr(l) && lp[l].use_hash && (RAW || (lp[l].hist <= W3_HIST_RAW && lp[l].align == 3u && lp[l].bits >= 6u))) {   // a bucketed map
                    const uint32_t claimers = (uint32_t)(__ballot(on && slot[l] == nullptr && ins[l] != 0xFFFFFFFFu) >> row0) & 0xFFFFu;
                    claim_rank[l] = (uint32_t)__popc(claimers & ((1u << r) - 1u));
                }
            }
            if (on) {
#pragma unroll
                for (int l = 0; l < NL; l++) {
                    if (!is_ctr(l)) continue;
                    const uint32_t nv = counter_update_packed(val[l], mybit);
                    if (slot[l]) { if (lp[l].use_hash) ds_st32(slot[l], nv); else pl_st32(slot[l], nv); }
                    else if (NM) {
                        // The kernels of large batches claim with ONE 8-byte compare-and-swap {context + 1, counts} at the slot the look-up found
                        // empty — no load first, no separate store of the counts (the claim was three dependent round trips on the nibble's
                        // critical path).  Bucketed maps: the path nodes of a nibble that claim are new contexts of ONE family, i.e. of one bucket,
                        // and a bucket fills from slot 0 without holes, so node k takes the k-th empty slot (claim_rank; 0 for a hashed map, where
                        // a probe ends at the first empty slot and none may be skipped) instead of racing for the first.  A slot that is taken
                        // after all (a full bucket continues in the next one; another context hashed there since the look-up) sends the claim on.
                        // Measured at 1e9 B: bench model 839 -> 1,012 - 1,024 MiB/s, Order0+1+2 1,019 -> 1,221 (profiles/r4_decode/decode_rates_claim_variants.txt).
                        uint64_t *tbl64 = reinterpret_cast<uint64_t *>(blk_tbl + lp[l].tbl_off);
                        const uint64_t mine = (uint64_t)(ctx[l] + 1u) | ((uint64_t)nv << 32);
                        uint32_t h = ins[l] + claim_rank[l];   // slots are numbered bucket * 16 + slot: consecutive, wrapping with the table
                        for (;;) {
                            h &= lp[l].hash_mask;
                            uint64_t expect = 0ull;
                            if (__hip_atomic_compare_exchange_strong(tbl64 + h, &expect, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                            h++;
                        }
                    }
                    else {   // the kernels of small batches (round-3 formats): probe again and claim; another path node of this nibble may be after
                             // the same empty slot.  (The one-CAS claim was measured here too: 11% SLOWER, 182 -> 163 MiB/s at 1e8 B, 565 -> 500 at 4e8.)
                        uint32_t *tbl = reinterpret_cast<uint32_t *>(blk_tbl + lp[l].tbl_off);
                        uint32_t h = (ctx[l] * 2654435761u) ^ (ctx[l] >> 15);
                        for (;;) {
                            h &= lp[l].hash_mask;
                            uint32_t key = ds_ld32(tbl + 2u * h);
                            if (key == 0u) {
                                uint32_t expect = 0u;
                                __hip_atomic_compare_exchange_strong(tbl + 2u * h, &expect, ctx[l] + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                key = expect == 0u ? ctx[l] + 1u : expect;
                            }
                            if (key == ctx[l] + 1u) { ds_st32(tbl + 2u * h + 1u, nv); break; }
                            h++;
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < NA; s++)
                    pl_st16(aslot[s], (uint16_t)(tv[s] + (((mybit ? 65535 : 0) - tv[s]) >> s_apm[s].rate)));   // arithmetic shift = floor
            }
            // With alignment_bits >= 3 the two nibbles of a byte have disjoint contexts (t mod 8 is part of them) and disjoint APM rows
            // (1..15 / 16..255): what this nibble stored can only be read two nibbles on — after the next nibble's loads have been waited
            // for, and gfx9's one counter for loads and stores then covers these stores too.  Only alignment 2 has to wait here.
            if (fence_each) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);
            }
            hist64 = (hist64 << D) | xk;
            t += (uint32_t)D;
            c0 = (c0 << D) | xk;
        }
        const uint32_t byte = c0 & 0xFFu;
        c1 = byte; c0 = 1u;
        if (!RAW && g.n_huff) hs.push_byte(g.huff, g.n_huff, byte);
        if (r == 0u) g.dout[off + i] = (uint8_t)byte;
    }
}

// what k_decode_spec covers (see the header comment)
static inline bool decode_spec_has_slot(const CmArgs &ca) {
    for (int l = 0; l < ca.g.n_leaves; l++)
        if (ca.g.leaf[l].kind == 1) return true;
    return false;
}
static inline bool decode_spec_covers(const CmArgs &ca) {
    if (ca.g.n_leaves < 1 || ca.g.n_leaves > 8 || ca.n_apm > 2) return false;
    for (int l = 0; l < ca.g.n_leaves; l++) {
        const LeafParam &lp = ca.g.leaf[l];
        if (lp.kind == 0 && !lp.frozen && lp.align < 2) return false;
    }
    return true;
}

// RAW: every leaf a live Counter table over raw history with alignment 3 (Order0 / Order1 / OrderN(.., 3) and their mixes, the bench model):
// the large-batch kernel is instantiated once more with all of that known at compile time and the leaf parameters read as scalars from the
// kernel arguments (the general kernel keeps eight leaves' parameters in LDS and branches on each leaf's kind, history and alignment per nibble)
static inline bool decode_spec_all_raw(const CmArgs &ca) {
    if (ca.dflags & 4u) return false;                           // W3_OPT_TUNE bit 19: the general kernel (tests run both)
    if (ca.g.n_leaves > 4 || ca.g.n_huff) return false;
    for (int l = 0; l < ca.g.n_leaves; l++) {
        const LeafParam &lp = ca.g.leaf[l];
        if (lp.kind != 0 || lp.frozen || lp.hist > W3_HIST_RAW || lp.align != 3u || lp.bits < 6u) return false;
    }
    return true;
}
// ... and the one shape with slot-state leaves that has an instance of its own: models.py's full_cm() — BASELINE configs[2], Order0 / Order1 /
// OrderN(27, 3) followed by the four slot-state orders (seven leaves, leaf-kind mask 0b1111000)
#define W3_DS_FULLCM_NL 7
#define W3_DS_FULLCM_KM 0x78
static inline bool decode_spec_full_cm_shape(const CmArgs &ca) {
    if ((ca.dflags & 4u) || ca.g.n_leaves != W3_DS_FULLCM_NL || ca.g.n_huff) return false;
    for (int l = 0; l < W3_DS_FULLCM_NL; l++) {
        const LeafParam &lp = ca.g.leaf[l];
        if ((W3_DS_FULLCM_KM >> l) & 1) { if (lp.kind != 1) return false; }
        else if (lp.kind != 0 || lp.frozen || lp.hist > W3_HIST_RAW || lp.align != 3u || lp.bits < 6u) return false;
    }
    return true;
}
template <bool NM>
static inline void launch_decode_spec_full_cm(const CmArgs &ca, uint32_t cnt, hipStream_t s) {
    const dim3 grid((cnt + 3u) / 4u), blk(64);
    switch (ca.n_apm) {
    case 0: hipLaunchKernelGGL((k_decode_spec<W3_DS_FULLCM_NL, 0, 4, true, NM, true, W3_DS_FULLCM_KM>), grid, blk, 0, s, ca); break;
    case 1: hipLaunchKernelGGL((k_decode_spec<W3_DS_FULLCM_NL, 1, 4, true, NM, true, W3_DS_FULLCM_KM>), grid, blk, 0, s, ca); break;
    default: hipLaunchKernelGGL((k_decode_spec<W3_DS_FULLCM_NL, 2, 4, true, NM, true, W3_DS_FULLCM_KM>), grid, blk, 0, s, ca); break;
    }
}

#define W3_DECODE_NM_MIN_BLOCKS_RAW 1024u   // ... for the all-raw-history instances (decode_spec_all_raw): never slower there (1e8 B 286 against 277 MiB/s, 4e8 B 823 against 677; equal below)
#define W3_DECODE_NM_MIN_BLOCKS 8192u   // batches of at least this many blocks decode with the nibble-major table formats (measured: 4e8 B 462 against 490 MiB/s, 1e9 B 885 against 684)

// up to eight leaves (the full CM has seven) and slot-state leaves: one instantiation with NL = 8, unused entries marked kind 2
template <bool HS, bool NM>
static inline void launch_decode_spec_8(CmArgs ca, uint32_t cnt, hipStream_t s) {
    for (int l = ca.g.n_leaves; l < 8; l++) { ca.g.leaf[l] = LeafParam{}; ca.g.leaf[l].kind = 2; ca.g.leaf[l].frozen = 1; }
    const dim3 grid((cnt + 3u) / 4u), blk(64);
    switch (ca.n_apm) {
    case 0: hipLaunchKernelGGL((k_decode_spec<8, 0, 4, HS, NM>), grid, blk, 0, s, ca); break;
    case 1: hipLaunchKernelGGL((k_decode_spec<8, 1, 4, HS, NM>), grid, blk, 0, s, ca); break;
    default: hipLaunchKernelGGL((k_decode_spec<8, 2, 4, HS, NM>), grid, blk, 0, s, ca); break;
    }
}

template <int NL, int D, bool NM, bool RAW = false>
static inline void launch_decode_spec_na(const CmArgs &ca, uint32_t cnt, hipStream_t s) {
    constexpr uint32_t BPW = 64u >> D;
    const dim3 grid((cnt + BPW - 1u) / BPW), blk(64);
    switch (ca.n_apm) {
    case 0: hipLaunchKernelGGL((k_decode_spec<NL, 0, D, false, NM, RAW>), grid, blk, 0, s, ca); break;
    case 1: hipLaunchKernelGGL((k_decode_spec<NL, 1, D, false, NM, RAW>), grid, blk, 0, s, ca); break;
    default: hipLaunchKernelGGL((k_decode_spec<NL, 2, D, false, NM, RAW>), grid, blk, 0, s, ca); break;
    }
}
// table formats of this batch (the APM tables' initialisation must agree: cm_run asks the same question)
static inline bool decode_spec_nibble_major(const CmArgs &ca, uint32_t cnt, int bits_per_group) {
    if (bits_per_group != 4) return false;                      // (the two-bit-group variant keeps the round-3 formats)
    if (ca.dflags & 1u) return false;                           // W3_OPT_TUNE bit 17: round-3 formats whatever the size
    if (ca.dflags & 2u) return true;                            // bit 18: nibble-major formats whatever the size
    return cnt >= (decode_spec_all_raw(ca) ? W3_DECODE_NM_MIN_BLOCKS_RAW : W3_DECODE_NM_MIN_BLOCKS);
}
// bits_per_group: 4 = the nibble (the shipped form), 2 = half a nibble (a tested variant)
static inline void launch_decode_spec(const CmArgs &ca, uint32_t cnt, hipStream_t s, int bits_per_group) {
    const bool nm = decode_spec_nibble_major(ca, cnt, bits_per_group);
    if (bits_per_group == 4 && decode_spec_full_cm_shape(ca)) { if (nm) launch_decode_spec_full_cm<true>(ca, cnt, s); else launch_decode_spec_full_cm<false>(ca, cnt, s); return; }
    if (decode_spec_has_slot(ca)) { if (nm) launch_decode_spec_8<true, true>(ca, cnt, s); else launch_decode_spec_8<true, false>(ca, cnt, s); return; }
    if (ca.g.n_leaves > 4) { if (nm) launch_decode_spec_8<false, true>(ca, cnt, s); else launch_decode_spec_8<false, false>(ca, cnt, s); return; }
    if (bits_per_group == 2) {
        switch (ca.g.n_leaves) {
        case 1: launch_decode_spec_na<1, 2, false>(ca, cnt, s); break;
        case 2: launch_decode_spec_na<2, 2, false>(ca, cnt, s); break;
        case 3: launch_decode_spec_na<3, 2, false>(ca, cnt, s); break;
        default: launch_decode_spec_na<4, 2, false>(ca, cnt, s); break;
        }
        return;
    }
    if (nm && decode_spec_all_raw(ca)) {
        switch (ca.g.n_leaves) {
        case 1: launch_decode_spec_na<1, 4, true, true>(ca, cnt, s); break;
        case 2: launch_decode_spec_na<2, 4, true, true>(ca, cnt, s); break;
        case 3: launch_decode_spec_na<3, 4, true, true>(ca, cnt, s); break;
        default: launch_decode_spec_na<4, 4, true, true>(ca, cnt, s); break;
        }
        return;
    }
    if (nm) {
        switch (ca.g.n_leaves) {
        case 1: launch_decode_spec_na<1, 4, true>(ca, cnt, s); break;
        case 2: launch_decode_spec_na<2, 4, true>(ca, cnt, s); break;
        case 3: launch_decode_spec_na<3, 4, true>(ca, cnt, s); break;
        default: launch_decode_spec_na<4, 4, true>(ca, cnt, s); break;
        }
        return;
    }
    if (decode_spec_all_raw(ca)) {
        switch (ca.g.n_leaves) {
        case 1: launch_decode_spec_na<1, 4, false, true>(ca, cnt, s); break;
        case 2: launch_decode_spec_na<2, 4, false, true>(ca, cnt, s); break;
        case 3: launch_decode_spec_na<3, 4, false, true>(ca, cnt, s); break;
        default: launch_decode_spec_na<4, 4, false, true>(ca, cnt, s); break;
        }
        return;
    }
    switch (ca.g.n_leaves) {
    case 1: launch_decode_spec_na<1, 4, false>(ca, cnt, s); break;
    case 2: launch_decode_spec_na<2, 4, false>(ca, cnt, s); break;
    case 3: launch_decode_spec_na<3, 4, false>(ca, cnt, s); break;
    default: launch_decode_spec_na<4, 4, false>(ca, cnt, s); break;
    }
}

}  // namespace w3
