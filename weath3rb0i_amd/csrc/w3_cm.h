// w3_cm.h — lane-per-block CONTEXT-MIXING kernel: the literal north-star design.  One wavefront lane
// owns one block's arithmetic coder and model state; the 12-bit state table (NaiveStateTable,
// state_table/naive.rs) and the APM stretch LUT are staged in LDS; slot-model contexts are resolved
// against a per-lane Cell/Slot hash map in HBM (hashmap.rs); Counter tables as in w3_generic.h.
//
// Runs every spec that contains a SLOT_STATE leaf or an APM stage, and every decode of such a spec
// (decode is serial by nature: the next context depends on the decoded bit, main.rs:131-140).
// The slot model, its replacement policy and the APM are BUILD-DEFINED (SURVEY §8 A19 ii-v): the
// reference has the primitives but no model that uses them.  Definitions: DESIGN.md §2.4.
#pragma once
#include "w3_generic.h"
#include "w3_tables.h"

namespace w3 {

struct ApmParam {
    uint8_t  ctx_kind;   // 0: row = c0 (partial byte, leading 1)   1: row = c0 | previous byte << 8
    uint8_t  rate;
    uint8_t  pad[6];
    uint64_t off;        // byte offset of this stage's u16[rows][33] table inside the lane's region
};

struct CmArgs {
    GenericArgs g;
    int n_apm;
    ApmParam apm[W3_MAX_APM];
    const uint2   *st;        // [kStSize] {prob | next0 << 16, next1 | conf << 16}
    const int16_t *stretch;   // [4096]
    const uint16_t *squash;   // [4095] (APM table initialisation only)
    uint32_t dflags;          // k_decode_spec table formats (W3_OPT_TUNE bits 17 / 18): bit 0 = the round-3 formats whatever the batch size, bit 1 = the nibble-major ones (default: by size)
};

// ---- Cell / Slot (hashmap.rs:31-129), byte-exact layout: 6 tag bytes + 90 state bytes ----------------
// Slot::get_idx (hashmap.rs:80-84): half-byte index of state (bit_id, nib_ctx) of slot `id`.
__device__ __forceinline__ uint32_t slot_idx(uint32_t id, uint32_t bit_id, uint32_t nib_ctx) {
    return (3u << bit_id) + 3u * nib_ctx + 45u * id - 3u;
}
__device__ __forceinline__ uint32_t slot_get_state(const uint8_t *cell, uint32_t id, uint32_t bit_id, uint32_t nib_ctx) {  // :86-97
    const uint32_t idx = slot_idx(id, bit_id, nib_ctx);
    const uint8_t *p = cell + 6u + (idx >> 1);
    const uint32_t v = ((uint32_t)p[0] << 8) | p[1];
    return (idx & 1u) ? (v & 0xFFFu) : (v >> 4);
}
__device__ __forceinline__ void slot_set_state(uint8_t *cell, uint32_t id, uint32_t bit_id, uint32_t nib_ctx, uint32_t ns) {  // :99-112
    const uint32_t idx = slot_idx(id, bit_id, nib_ctx);
    uint8_t *p = cell + 6u + (idx >> 1);
    if (!(idx & 1u)) {
        p[0] = (uint8_t)(ns >> 4);
        p[1] = (uint8_t)(((ns << 4) & 0xF0u) | (p[1] & 0x0Fu));
    } else {
        p[0] = (uint8_t)((ns >> 8) | (p[0] & 0xF0u));
        p[1] = (uint8_t)ns;
    }
}

// splitmix64 finaliser over (order, previous `order` bytes, nibble marker) — build-defined context hash
__device__ __forceinline__ uint64_t slot_hash(uint32_t order, uint64_t hist_bytes, bool second, uint32_t hi_nib) {
    uint64_t k = order ? (hist_bytes & ((1ull << (8u * order)) - 1ull)) : 0ull;
    k = (k << 8) | (second ? (0x10u | hi_nib) : 0u);
    k += (uint64_t)(order + 1u) * 0x9E3779B97F4A7C15ull;
    k ^= k >> 30; k *= 0xBF58476D1CE4E5B9ull;
    k ^= k >> 27; k *= 0x94D049BB133111EBull;
    k ^= k >> 31;
    return k;
}

// HashMap::get_slot + Cell::get_slot (hashmap.rs:25-28, 42-63): cell by the HIGH hash bits, 12-bit tags
// compared in the order id 3, 2, 1, 0.  Miss (the reference's TODO, :64-68): victim = fewest observations
// in the slot's first-bit state, candidates in the order 1, 0, 2, 3; tag stored, 15 states cleared.
__device__ __forceinline__ uint8_t *slot_select(uint8_t *cells, uint32_t log_cells, uint64_t h, const uint2 *s_st, uint32_t &id_out) {
    uint8_t *cell = cells + (h >> (64u - log_cells)) * 96ull;
    const uint32_t tag = (uint32_t)h & 0xFFFu;
    uint64_t hc = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) hc = (hc << 8) | cell[i];
    int id = -1;
    if (tag == (uint32_t)(hc & 0xFFFu)) id = 3;
    else if (tag == (uint32_t)((hc >> 12) & 0xFFFu)) id = 2;
    else if (tag == (uint32_t)((hc >> 24) & 0xFFFu)) id = 1;
    else if (tag == (uint32_t)((hc >> 36) & 0xFFFu)) id = 0;
    if (id < 0) {
        uint32_t best = 0xFFFFFFFFu;
        const int cand[4] = {1, 0, 2, 3};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t conf = s_st[slot_get_state(cell, (uint32_t)cand[k], 0u, 0u)].y >> 16;
            if (conf < best) { best = conf; id = cand[k]; }
        }
        const uint32_t sh = 12u * (3u - (uint32_t)id);
        hc = (hc & ~(0xFFFull << sh)) | ((uint64_t)tag << sh);
#pragma unroll
        for (int i = 5; i >= 0; i--) { cell[i] = (uint8_t)hc; hc >>= 8; }
        for (uint32_t bit_id = 0; bit_id < 4u; bit_id++)
            for (uint32_t c = 0; c < (1u << bit_id); c++) slot_set_state(cell, (uint32_t)id, bit_id, c, 0u);
    }
    id_out = (uint32_t)id;
    return cell;
}

// APM tables start as the identity map: t[row][j] = squash((j - 16) * 128)
// rows x 33 identity entries per lane.  nibble_major: k_decode_spec's layout [page][17 groups][33][16 nodes] (rows = pages * 272): an entry's
// value depends on its j alone, which is (index / 16) % 33 there and index % 33 in the row-major tables of every other kernel.
__global__ void __launch_bounds__(256) k_cm_init_apm(uint8_t *tables, uint64_t lane_stride, uint64_t off, uint32_t rows,
                                                    uint32_t n_lanes, const uint16_t *squash, uint32_t nibble_major) {
    const uint64_t per_lane = (uint64_t)rows * 33u;
    const uint64_t total = per_lane * n_lanes;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t lane = i / per_lane;
        const uint64_t e = i - lane * per_lane;
        const uint32_t j = (uint32_t)((nibble_major ? e / 16u : e) % 33u);
        int d = ((int)j - 16) * 128;
        d = d < -2047 ? -2047 : d > 2047 ? 2047 : d;
        reinterpret_cast<uint16_t *>(tables + lane * lane_stride + off)[e] = squash[d + 2047];
    }
}

template <bool DECODE>
__global__ void __launch_bounds__(64) k_cm(CmArgs a) {
    __shared__ uint2   s_st[kStSize];
    __shared__ int16_t s_str[4096];
    for (uint32_t i = threadIdx.x; i < (uint32_t)kStSize; i += 64u) s_st[i] = a.st[i];
    for (uint32_t i = threadIdx.x; i < 4096u; i += 64u) s_str[i] = a.stretch[i];
    __shared__ LeafParam s_leaf[W3_MAX_LEAVES];
    __shared__ ApmParam s_apm[W3_MAX_APM];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < W3_MAX_APM; k++) s_apm[k] = a.apm[k];
    }
    stage_leaves(s_leaf, a.g);   // (also the barrier for the tables above)

    const GenericArgs &g = a.g;
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= g.n_lanes) return;
    const uint32_t b = g.first_block + lane;
    const uint64_t off = (uint64_t)b * g.block_size;
    const uint32_t len = (uint32_t)((g.n - off) < g.block_size ? (g.n - off) : g.block_size);
    uint8_t *lane_tbl = g.tables + (uint64_t)lane * g.lane_stride;

    Encoder enc; Decoder dec;
    if (DECODE) dec.init(g.cin + g.coffs[b], g.clens[b]);
    else enc.init(g.stripes + (uint64_t)lane * g.stripe_cap, g.stripe_cap);

    // slot leaves: the cell/slot of the current nibble (selected for an empty history at construction)
    uint8_t *cellp[W3_MAX_LEAVES]; uint32_t sid[W3_MAX_LEAVES];
    for (int l = 0; l < g.n_leaves; l++) {
        cellp[l] = nullptr; sid[l] = 0;
        if (s_leaf[l].kind == 1) cellp[l] = slot_select(lane_tbl + s_leaf[l].tbl_off, s_leaf[l].log_cells, slot_hash(s_leaf[l].order, 0ull, false, 0u), s_st, sid[l]);
    }

    uint64_t hist64 = 0, hist_bytes = 0; uint32_t t = 0, c0 = 1, c1 = 0;
    HuffState hs;
    for (uint32_t i = 0; i < len; i++) {
        uint32_t byte = DECODE ? 0u : g.in[off + i];
        for (int s = 7; s >= 0; s--) {
            const uint32_t bit_id = (uint32_t)(7 - s) & 3u, nib_ctx = c0 & ((1u << bit_id) - 1u);
            // predict: leftmost leaf of maximal |p - 1/2| (BestOfTwo tree, models/mod.rs:67-69)
            uint32_t p = 32768u, best = 0u; bool first = true;
            uint32_t *cslot[W3_MAX_LEAVES]; uint32_t st[W3_MAX_LEAVES];
            for (int l = 0; l < g.n_leaves; l++) {
                const LeafParam &lp = s_leaf[l];
                uint32_t pl = 32768u;
                cslot[l] = nullptr; st[l] = 0;
                if (lp.kind == 1) {
                    st[l] = slot_get_state(cellp[l], sid[l], bit_id, nib_ctx);
                    pl = s_st[st[l]].x & 0xFFFFu;                       // StateTable::p  state_table/mod.rs:47-49
                } else if (!lp.frozen) {
                    cslot[l] = leaf_slot(lp, lane_tbl, leaf_ctx(lp, hist64, t, hs, a.g.huff));
                    pl = counter_p_packed(*cslot[l]);
                }
                const uint32_t d = opinion_dist(pl);
                if (first || d > best) { p = pl; best = d; first = false; }
            }
            // APM chain (build-defined): refine p through each stage
            uint16_t *aslot[W3_MAX_APM];
            for (int k = 0; k < a.n_apm; k++) {
                const uint32_t row = s_apm[k].ctx_kind ? (c0 | (c1 << 8)) : c0;
                const uint32_t pos = (uint32_t)((int)s_str[p >> 4] + 2048) * 32u;
                const uint32_t j = pos >> 12, w = pos & 4095u;
                uint16_t *tr = reinterpret_cast<uint16_t *>(lane_tbl + s_apm[k].off) + row * 33u + j;
                const uint32_t pa = ((uint32_t)tr[0] * (4096u - w) + (uint32_t)tr[1] * w) >> 12;
                aslot[k] = tr + (w >> 11);
                uint32_t o = (p + 3u * pa + 2u) >> 2;
                p = o < 1u ? 1u : o > 65535u ? 65535u : o;
            }
            uint32_t bit;
            if (DECODE) { bit = dec.decode(p); byte = (byte << 1) | bit; }
            else bit = (byte >> s) & 1u;
            // Model::update = adapt (train the current context) then advance  models/mod.rs:28-31
            for (int k = 0; k < a.n_apm; k++) {
                const int tv = (int)*aslot[k];
                *aslot[k] = (uint16_t)(tv + (((bit ? 65535 : 0) - tv) >> s_apm[k].rate));   // arithmetic shift = floor
            }
            for (int l = 0; l < g.n_leaves; l++) {
                if (s_leaf[l].kind == 1) slot_set_state(cellp[l], sid[l], bit_id, nib_ctx, bit ? (s_st[st[l]].y & 0xFFFFu) : (s_st[st[l]].x >> 16));
                else if (cslot[l]) *cslot[l] = counter_update_packed(*cslot[l], bit);
            }
            hist64 = (hist64 << 1) | bit;
            t++;
            c0 = (c0 << 1) | bit;
            if (!DECODE) enc.encode(bit, p);
            if (bit_id == 3u) {   // nibble boundary: one cell touch per nibble (hashslots.md:3-4)
                const bool second = c0 < 256u;
                if (!second) { hist_bytes = (hist_bytes << 8) | (c0 & 0xFFu); c1 = c0 & 0xFFu; c0 = 1u; }
                if (i + 1u < len || second)
                    for (int l = 0; l < g.n_leaves; l++)
                        if (s_leaf[l].kind == 1)
                            cellp[l] = slot_select(lane_tbl + s_leaf[l].tbl_off, s_leaf[l].log_cells,
                                                   slot_hash(s_leaf[l].order, hist_bytes, second, c0 & 15u), s_st, sid[l]);
            }
        }
        if (g.n_huff) hs.push_byte(g.huff, g.n_huff, byte & 0xFFu);
        if (DECODE) g.dout[off + i] = (uint8_t)byte;
    }
    if (!DECODE) {
        if (g.out_bits) g.out_bits[b] = enc.stats_bits();
        const uint32_t produced = enc.flush();
        g.out_len[b] = produced;
        if (produced > g.stripe_cap) atomicOr(g.overflow, 1u);
    }
}

// ---------------------------------------------------------------------------
// k_cm with the slot-state leaves' current Cell STAGED IN LDS (up to 8 slot leaves): one 96-byte load when a nibble's
// cell is selected, byte-exact operations on the staged copy during the nibble's four steps, one 96-byte write-back when
// the leaf moves on.  k_cm itself touches global memory four times per leaf and step (two byte loads + a two-byte
// read-modify-write), each a dependent ~1 us round trip for a lone lane: the full CM decoded at 14 MiB/s.
// ---------------------------------------------------------------------------
typedef uint32_t cm_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint32_t cm_lds_u32;
#define W3_CM_STAGED_MAX 8

// dword d of the lane's staged cell is at cb[(d >> 2) * 256 + (d & 3)] (cb = &s_cell[k][0][lane] viewed as dwords)
__device__ __forceinline__ uint32_t cmc_off(uint32_t d) { return (d >> 2) * 256u + (d & 3u); }
struct CmCellRef { uint32_t olo, ohi, lo, hi, sh; };
__device__ __forceinline__ uint32_t cmc_get16(cm_lds_u32 *cb, uint32_t byte, CmCellRef &r) {   // big-endian u16 at `byte` (hashmap.rs:86-97)
    const uint32_t d = byte >> 2;
    r.olo = cmc_off(d);
    r.ohi = cmc_off(d == 23u ? 22u : d + 1u);   // byte <= 94: the pair never spills out of dword 23; 22 is a harmless stand-in
    r.lo = cb[r.olo]; r.hi = cb[r.ohi];
    r.sh = 8u * (byte & 3u);
    const uint32_t q = (uint32_t)((((uint64_t)r.hi << 32) | r.lo) >> r.sh);
    return ((q & 0xFFu) << 8) | ((q >> 8) & 0xFFu);
}
__device__ __forceinline__ void cmc_put16(cm_lds_u32 *cb, const CmCellRef &r, uint32_t v) {
    const uint64_t x = ((v >> 8) & 0xFFu) | ((v & 0xFFu) << 8);
    const uint64_t Q = (((((uint64_t)r.hi << 32) | r.lo)) & ~(0xFFFFull << r.sh)) | (x << r.sh);
    cb[r.olo] = (uint32_t)Q;
    cb[r.ohi] = (uint32_t)(Q >> 32);   // unchanged unless the pair straddles two dwords
}
__device__ __forceinline__ uint32_t cmc_state(cm_lds_u32 *cb, uint32_t idx, CmCellRef &r, uint32_t &v) {
    v = cmc_get16(cb, 6u + (idx >> 1), r);
    return (idx & 1u) ? (v & 0xFFFu) : (v >> 4);
}
__device__ __forceinline__ void cmc_set_state(cm_lds_u32 *cb, uint32_t idx, const CmCellRef &r, uint32_t v, uint32_t ns) {
    cmc_put16(cb, r, (idx & 1u) ? ((v & 0xF000u) | ns) : ((ns << 4) | (v & 0xFu)));
}
// Cell::get_slot on the staged cell + the replacement policy of slot_select above
__device__ __forceinline__ uint32_t cmc_select(cm_lds_u32 *cb, uint32_t tag, const uint2 *s_st) {
    const uint32_t tw0 = cb[cmc_off(0)], tw1 = cb[cmc_off(1)];
    const uint64_t hc = ((uint64_t)__builtin_bswap32(tw0) << 16) | (__builtin_bswap32(tw1) >> 16);
    int id = -1;
    if (tag == (uint32_t)(hc & 0xFFFu)) id = 3;
    else if (tag == (uint32_t)((hc >> 12) & 0xFFFu)) id = 2;
    else if (tag == (uint32_t)((hc >> 24) & 0xFFFu)) id = 1;
    else if (tag == (uint32_t)((hc >> 36) & 0xFFFu)) id = 0;
    if (id < 0) {
        uint32_t best = 0xFFFFFFFFu;
        const int cand[4] = {1, 0, 2, 3};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            CmCellRef r; uint32_t v;
            const uint32_t conf = s_st[cmc_state(cb, 45u * (uint32_t)cand[k], r, v)].y >> 16;
            if (conf < best) { best = conf; id = cand[k]; }
        }
        const uint32_t shv = 12u * (3u - (uint32_t)id);
        const uint64_t hc2 = (hc & ~(0xFFFull << shv)) | ((uint64_t)tag << shv);
        __asm__ volatile("" ::: "memory");
        cb[cmc_off(0)] = __builtin_bswap32((uint32_t)(hc2 >> 16));
        cb[cmc_off(1)] = (tw1 & 0xFFFF0000u) | (__builtin_bswap32((uint32_t)(hc2 << 16)) & 0x0000FFFFu);
        __asm__ volatile("" ::: "memory");
        for (uint32_t bit_id = 0; bit_id < 4u; bit_id++)
            for (uint32_t c = 0; c < (1u << bit_id); c++) {
                CmCellRef r; uint32_t v;
                const uint32_t idx = slot_idx((uint32_t)id, bit_id, c);
                (void)cmc_state(cb, idx, r, v);
                cmc_set_state(cb, idx, r, v, 0u);
                __asm__ volatile("" ::: "memory");
            }
    }
    return (uint32_t)id;
}

template <bool DECODE>
__global__ void __launch_bounds__(64) k_cm_staged(CmArgs a) {
    __shared__ uint2   s_st[kStSize];
    __shared__ int16_t s_str[4096];
    __shared__ cm_u32x4 s_cell[W3_CM_STAGED_MAX][6][64];
    for (uint32_t i = threadIdx.x; i < (uint32_t)kStSize; i += 64u) s_st[i] = a.st[i];
    for (uint32_t i = threadIdx.x; i < 4096u; i += 64u) s_str[i] = a.stretch[i];
    __shared__ LeafParam s_leaf[W3_MAX_LEAVES];
    __shared__ ApmParam s_apm[W3_MAX_APM];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < W3_MAX_APM; k++) s_apm[k] = a.apm[k];
    }
    stage_leaves(s_leaf, a.g);   // (also the barrier for the tables above)

    const GenericArgs &g = a.g;
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= g.n_lanes) return;
    const uint32_t b = g.first_block + lane;
    const uint64_t off = (uint64_t)b * g.block_size;
    const uint32_t len = (uint32_t)((g.n - off) < g.block_size ? (g.n - off) : g.block_size);
    uint8_t *lane_tbl = g.tables + (uint64_t)lane * g.lane_stride;

    Encoder enc; Decoder dec;
    if (DECODE) dec.init(g.cin + g.coffs[b], g.clens[b]);
    else enc.init(g.stripes + (uint64_t)lane * g.stripe_cap, g.stripe_cap);

    // slot leaf l uses staging area sk[l]; cellp[l] = the global address of the staged cell
    uint8_t *cellp[W3_MAX_LEAVES]; uint32_t sid[W3_MAX_LEAVES], sk[W3_MAX_LEAVES];
    auto stage = [&](int l, uint64_t h) {
        uint8_t *cell = lane_tbl + s_leaf[l].tbl_off + (h >> (64u - s_leaf[l].log_cells)) * 96ull;
        cm_lds_u32 *cb = (cm_lds_u32 *)&s_cell[sk[l]][0][threadIdx.x];
        if (cell != cellp[l]) {
            __asm__ volatile("" ::: "memory");
            if (cellp[l]) {   // write the previous cell back
                cm_u32x4 *dst = reinterpret_cast<cm_u32x4 *>(cellp[l]);
#pragma unroll
                for (int q = 0; q < 6; q++) dst[q] = s_cell[sk[l]][q][threadIdx.x];
            }
            const cm_u32x4 *src = reinterpret_cast<const cm_u32x4 *>(cell);
#pragma unroll
            for (int q = 0; q < 6; q++) s_cell[sk[l]][q][threadIdx.x] = src[q];
            __asm__ volatile("" ::: "memory");
            cellp[l] = cell;
        }
        sid[l] = cmc_select(cb, (uint32_t)h & 0xFFFu, s_st);
    };
    {
        uint32_t nk = 0;
        for (int l = 0; l < g.n_leaves; l++) {
            cellp[l] = nullptr; sid[l] = 0; sk[l] = 0;
            if (s_leaf[l].kind == 1) { sk[l] = nk++; stage(l, slot_hash(s_leaf[l].order, 0ull, false, 0u)); }
        }
    }

    uint64_t hist64 = 0, hist_bytes = 0; uint32_t t = 0, c0 = 1, c1 = 0;
    HuffState hs;
    for (uint32_t i = 0; i < len; i++) {
        uint32_t byte = DECODE ? 0u : g.in[off + i];
        for (int s = 7; s >= 0; s--) {
            const uint32_t bit_id = (uint32_t)(7 - s) & 3u, nib_ctx = c0 & ((1u << bit_id) - 1u);
            // predict: leftmost leaf of maximal |p - 1/2| (BestOfTwo tree, models/mod.rs:67-69)
            uint32_t p = 32768u, best = 0u; bool first = true;
            uint32_t *cslot[W3_MAX_LEAVES]; uint32_t st[W3_MAX_LEAVES];
            for (int l = 0; l < g.n_leaves; l++) {
                const LeafParam &lp = s_leaf[l];
                uint32_t pl = 32768u;
                cslot[l] = nullptr; st[l] = 0;
                if (lp.kind == 1) {
                    cm_lds_u32 *cb = (cm_lds_u32 *)&s_cell[sk[l]][0][threadIdx.x];
                    CmCellRef r; uint32_t v;
                    st[l] = cmc_state(cb, slot_idx(sid[l], bit_id, nib_ctx), r, v);
                    pl = s_st[st[l]].x & 0xFFFFu;                       // StateTable::p  state_table/mod.rs:47-49
                } else if (!lp.frozen) {
                    cslot[l] = leaf_slot(lp, lane_tbl, leaf_ctx(lp, hist64, t, hs, a.g.huff));
                    pl = counter_p_packed(*cslot[l]);
                }
                const uint32_t d = opinion_dist(pl);
                if (first || d > best) { p = pl; best = d; first = false; }
            }
            // APM chain (build-defined): refine p through each stage
            uint16_t *aslot[W3_MAX_APM];
            for (int k = 0; k < a.n_apm; k++) {
                const uint32_t row = s_apm[k].ctx_kind ? (c0 | (c1 << 8)) : c0;
                const uint32_t pos = (uint32_t)((int)s_str[p >> 4] + 2048) * 32u;
                const uint32_t j = pos >> 12, w = pos & 4095u;
                uint16_t *tr = reinterpret_cast<uint16_t *>(lane_tbl + s_apm[k].off) + row * 33u + j;
                const uint32_t pa = ((uint32_t)tr[0] * (4096u - w) + (uint32_t)tr[1] * w) >> 12;
                aslot[k] = tr + (w >> 11);
                uint32_t o = (p + 3u * pa + 2u) >> 2;
                p = o < 1u ? 1u : o > 65535u ? 65535u : o;
            }
            uint32_t bit;
            if (DECODE) { bit = dec.decode(p); byte = (byte << 1) | bit; }
            else bit = (byte >> s) & 1u;
            // Model::update = adapt (train the current context) then advance  models/mod.rs:28-31
            for (int k = 0; k < a.n_apm; k++) {
                const int tv = (int)*aslot[k];
                *aslot[k] = (uint16_t)(tv + (((bit ? 65535 : 0) - tv) >> s_apm[k].rate));   // arithmetic shift = floor
            }
            for (int l = 0; l < g.n_leaves; l++) {
                if (s_leaf[l].kind == 1) {
                    cm_lds_u32 *cb = (cm_lds_u32 *)&s_cell[sk[l]][0][threadIdx.x];
                    CmCellRef r; uint32_t v;   // (re-read: keeping the references of up to 16 leaves across the decode went to scratch)
                    const uint32_t idx = slot_idx(sid[l], bit_id, nib_ctx);
                    (void)cmc_state(cb, idx, r, v);
                    cmc_set_state(cb, idx, r, v, bit ? (s_st[st[l]].y & 0xFFFFu) : (s_st[st[l]].x >> 16));
                    __asm__ volatile("" ::: "memory");
                } else if (cslot[l]) *cslot[l] = counter_update_packed(*cslot[l], bit);
            }
            hist64 = (hist64 << 1) | bit;
            t++;
            c0 = (c0 << 1) | bit;
            if (!DECODE) enc.encode(bit, p);
            if (bit_id == 3u) {   // nibble boundary: one cell touch per nibble (hashslots.md:3-4)
                const bool second = c0 < 256u;
                if (!second) { hist_bytes = (hist_bytes << 8) | (c0 & 0xFFu); c1 = c0 & 0xFFu; c0 = 1u; }
                if (i + 1u < len || second)
                    for (int l = 0; l < g.n_leaves; l++)
                        if (s_leaf[l].kind == 1) stage(l, slot_hash(s_leaf[l].order, hist_bytes, second, c0 & 15u));
            }
        }
        if (g.n_huff) hs.push_byte(g.huff, g.n_huff, byte & 0xFFu);
        if (DECODE) g.dout[off + i] = (uint8_t)byte;
    }
    if (!DECODE) {
        if (g.out_bits) g.out_bits[b] = enc.stats_bits();
        const uint32_t produced = enc.flush();
        g.out_len[b] = produced;
        if (produced > g.stripe_cap) atomicOr(g.overflow, 1u);
    }
}

// ---------------------------------------------------------------------------
// Counter leaves (1..4) + APM chain, no slot-state leaves: k_generic_nl (all leaves' Counter loads of a step in flight
// together, w3_generic.h) followed by the APM stages.  The decoder of the bench's default model.
// ---------------------------------------------------------------------------
template <bool DECODE, int NL>
__global__ void __launch_bounds__(64) k_cm_nl(CmArgs a) {
    __shared__ int16_t s_str[4096];
    for (uint32_t i = threadIdx.x; i < 4096u; i += 64u) s_str[i] = a.stretch[i];
    __shared__ LeafParam s_leaf[W3_MAX_LEAVES];
    __shared__ ApmParam s_apm[W3_MAX_APM];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < W3_MAX_APM; k++) s_apm[k] = a.apm[k];
    }
    stage_leaves(s_leaf, a.g);   // (also the barrier for the tables above)
    const GenericArgs &g = a.g;
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= g.n_lanes) return;
    const uint32_t b = g.first_block + lane;
    const uint64_t off = (uint64_t)b * g.block_size;
    const uint32_t len = (uint32_t)((g.n - off) < g.block_size ? (g.n - off) : g.block_size);
    uint8_t *lane_tbl = g.tables + (uint64_t)lane * g.lane_stride;
    const LeafParam *lp = s_leaf;
    Encoder enc; Decoder dec;
    if (DECODE) dec.init(g.cin + g.coffs[b], g.clens[b]);
    else enc.init(g.stripes + (uint64_t)lane * g.stripe_cap, g.stripe_cap);

    uint64_t hist64 = 0; uint32_t t = 0, c0 = 1, c1 = 0;
    HuffState hs;
    for (uint32_t i = 0; i < len; i++) {
        uint32_t byte = DECODE ? 0u : g.in[off + i];
        for (int s = 7; s >= 0; s--) {
            uint32_t *slot[NL]; uint32_t val[NL], key[NL], ctx[NL];
#pragma unroll
            for (int l = 0; l < NL; l++) {
                slot[l] = nullptr; val[l] = 0u; key[l] = 0u; ctx[l] = 0u;
                if (!lp[l].frozen) {
                    ctx[l] = leaf_ctx(lp[l], hist64, t, hs, a.g.huff);
                    uint32_t *tbl = reinterpret_cast<uint32_t *>(lane_tbl + lp[l].tbl_off);
                    slot[l] = lp[l].use_hash ? tbl + 2u * (((ctx[l] * 2654435761u) ^ (ctx[l] >> 15)) & lp[l].hash_mask) : tbl + ctx[l];
                }
            }
#pragma unroll
            for (int l = 0; l < NL; l++) {
                if (slot[l]) {
                    if (!lp[l].use_hash) val[l] = *slot[l];
                    else { const uint2 kv = *reinterpret_cast<const uint2 *>(slot[l]); key[l] = kv.x; val[l] = kv.y; }
                }
            }
            uint32_t p = 32768u, best = 0u;
#pragma unroll
            for (int l = 0; l < NL; l++) {
                if (slot[l] && lp[l].use_hash) {   // leaf_slot semantics: hit, claim an empty slot, or keep probing
                    if (key[l] == ctx[l] + 1u) slot[l] += 1;
                    else if (key[l] == 0u) { slot[l][0] = ctx[l] + 1u; slot[l] += 1; val[l] = 0u; }
                    else { slot[l] = leaf_slot(lp[l], lane_tbl, ctx[l]); val[l] = *slot[l]; }
                }
                const uint32_t pl = slot[l] ? counter_p_packed(val[l]) : 32768u;
                const uint32_t d = opinion_dist(pl);
                if (l == 0 || d > best) { p = pl; best = d; }
            }
            // APM chain (build-defined)
            uint16_t *aslot[W3_MAX_APM];
            for (int k = 0; k < a.n_apm; k++) {
                const uint32_t row = s_apm[k].ctx_kind ? (c0 | (c1 << 8)) : c0;
                const uint32_t pos = (uint32_t)((int)s_str[p >> 4] + 2048) * 32u;
                const uint32_t j = pos >> 12, w = pos & 4095u;
                uint16_t *tr = reinterpret_cast<uint16_t *>(lane_tbl + s_apm[k].off) + row * 33u + j;
                const uint32_t pa = ((uint32_t)tr[0] * (4096u - w) + (uint32_t)tr[1] * w) >> 12;
                aslot[k] = tr + (w >> 11);
                uint32_t o = (p + 3u * pa + 2u) >> 2;
                p = o < 1u ? 1u : o > 65535u ? 65535u : o;
            }
            uint32_t bit;
            if (DECODE) { bit = dec.decode(p); byte = (byte << 1) | bit; }
            else bit = (byte >> s) & 1u;
            for (int k = 0; k < a.n_apm; k++) {
                const int tv = (int)*aslot[k];
                *aslot[k] = (uint16_t)(tv + (((bit ? 65535 : 0) - tv) >> s_apm[k].rate));   // arithmetic shift = floor
            }
#pragma unroll
            for (int l = 0; l < NL; l++)
                if (slot[l]) *slot[l] = counter_update_packed(val[l], bit);
            hist64 = (hist64 << 1) | bit;
            t++;
            c0 = (c0 << 1) | bit;
            if (c0 >= 256u) { c1 = c0 & 0xFFu; c0 = 1u; }
            if (!DECODE) enc.encode(bit, p);
        }
        if (g.n_huff) hs.push_byte(g.huff, g.n_huff, byte & 0xFFu);
        if (DECODE) g.dout[off + i] = (uint8_t)byte;
    }
    if (!DECODE) {
        if (g.out_bits) g.out_bits[b] = enc.stats_bits();
        const uint32_t produced = enc.flush();
        g.out_len[b] = produced;
        if (produced > g.stripe_cap) atomicOr(g.overflow, 1u);
    }
}

}  // namespace w3
