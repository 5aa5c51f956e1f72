// w3_generic.h — GENERIC device path: one wavefront lane runs the reference's
// bit loop (main.rs:103-109 encode, :131-140 decode) for one block, with its
// model state (Counter tables) in HBM.  Handles every model spec, including
// the ones the two-phase fast path does not cover, and is the only decoder
// (decoding is serial by nature: the next context depends on the decoded bit).
#pragma once
#include "w3_device.h"
#include "../../include/w3hip.h"

namespace w3 {

struct LeafParam {
    uint8_t  bits, align, hist, max_bits, frozen, use_hash;
    uint8_t  kind;           // 0 = Counter table leaf, 1 = slot-state leaf (w3_cm.h)
    uint8_t  log_cells;      // slot leaf: HashMap log_cell_count
    uint8_t  order;          // slot leaf: previous bytes in the context
    uint8_t  huff_idx;       // HuffHistory leaf: its table set in GenericArgs::huff
    uint16_t table[8];       // StationaryModel table (ACHistory)
    uint64_t tbl_off;        // byte offset of this leaf's table inside the lane's region
    uint32_t hash_mask;      // slots-1 when use_hash
    uint32_t hist_mask;      // (1<<(bits-align))-1
    const uint4 *lut;        // ACHistory leaves: [65536][8] coder states after the 16 most recent history bits (k_achash_lut), or null
};

struct GenericArgs {
    int       n_leaves;
    LeafParam leaf[W3_MAX_LEAVES];
    // geometry
    uint64_t  n;             // total original bytes
    uint32_t  block_size;
    uint32_t  first_block, n_lanes;
    // model tables: lane l owns [tables + l*lane_stride, +lane_stride)
    uint8_t  *tables;
    uint64_t  lane_stride;
    // encode
    const uint8_t *in;       // original bytes
    uint8_t  *stripes;       // lane-major output stripes (encode)
    uint32_t  stripe_cap;
    uint32_t *out_len;       // [nblocks] bytes produced
    uint32_t *out_bits;      // [nblocks] ACStats bit count of each block (helpers.rs:60-90), or null
    uint32_t *overflow;      // set to 1 if any stripe overflowed
    // decode
    const uint8_t  *cin;     // concatenated block streams
    const uint64_t *coffs;   // [nblocks] offsets into cin
    const uint32_t *clens;   // [nblocks]
    uint8_t  *dout;          // original bytes out
    // HuffHistory leaves: the spec's table sets, copied to the device for the call
    const w3_huff_table *huff;
    int       n_huff;
};

// Context of a leaf at step t, given the lane's common 64-bit history (newest
// bit at bit 0) — OrderN::update (models/ordern.rs:35-43) and
// OrderNEntropy::update (models/ordern_entropy.rs:36-45) in closed form:
//   history  = last (bits-align) input bits      (zeros before the block start)
//   alignment= t mod 2^align                     (incremented once per update)
// Both models start with ctx = 0 and only form ctx inside update(), so t == 0
// is ctx 0 even for ACHistory (whose hash of an empty history is not 0).
// HuffHistory (history/huff_history.rs:58-76) keeps one piece of running state: compressed_bits, the concatenated codes of
// the completed bytes (u32: old bits fall off the top).  One value per table set, advanced by the kernels' loops once per
// completed byte — hash() does it at alignment 0, i.e. before the first context of the next byte is formed.
struct HuffState {
    uint32_t cb0 = 0u, cb1 = 0u, cb2 = 0u, cb3 = 0u;
    __device__ __forceinline__ uint32_t get(uint32_t k) const { return k == 0u ? cb0 : k == 1u ? cb1 : k == 2u ? cb2 : cb3; }
    __device__ __forceinline__ void push_byte(const w3_huff_table *tb, int n, uint32_t byte) {
        if (n > 0) cb0 = (cb0 << tb[0].len[byte]) | tb[0].code[byte];
        if (n > 1) cb1 = (cb1 << tb[1].len[byte]) | tb[1].code[byte];
        if (n > 2) cb2 = (cb2 << tb[2].len[byte]) | tb[2].code[byte];
        if (n > 3) cb3 = (cb3 << tb[3].len[byte]) | tb[3].code[byte];
    }
};

__device__ __forceinline__ uint32_t leaf_ctx(const LeafParam &lp, uint64_t hist64, uint32_t t, const HuffState &hs, const w3_huff_table *huff) {
    if (t == 0u) return 0u;
    uint32_t h;
    if (lp.hist == W3_HIST_HUFF) {
        const uint32_t al = t & 7u;
        const uint32_t rem = ((uint32_t)hist64 & ((1u << al) - 1u)) | (1u << al);   // the partial byte with a leading 1 (:71-73)
        const w3_huff_table &tb = huff[lp.huff_idx];
        h = (hs.get(lp.huff_idx) << tb.rem_len[rem]) | tb.rem_code[rem];            // :74-75
    } else if (lp.hist == 2) {
        if (lp.lut) {
            // the 16-bit prefix table of k_achash (w3_predict.h): most hashes are complete there; the rest resume at step 16.
            // (The literal one-bit-at-a-time form made the reference's default model decode at 12 MiB/s.)
            const uint32_t j = t & 7u;
            const uint4 e = lp.lut[((uint32_t)hist64 & 0xFFFFu) * 8u + j];
            ACHashState st; st.x1 = e.x; st.x2 = e.y; st.hash = e.z; st.meta = e.w;
            if (!(e.w >> 31)) {
                uint32_t rot[8];
#pragma unroll
                for (int r = 0; r < 8; r++) { const uint32_t p = lp.table[(j + 7u - (uint32_t)r) & 7u]; rot[r] = p ? (p << 16) : 1u; }
                st = ac_history_hash_steps(hist64, lp.max_bits, rot, st, 16, 64);
            }
            h = ac_hash_finish(st, lp.max_bits);
        } else h = ac_history_hash(hist64, t, lp.max_bits, lp.table);
    } else h = (uint32_t)hist64;
    h &= lp.hist_mask;
    return (h << lp.align) | (t & ((1u << lp.align) - 1u));
}

// Exact-keyed counter slot for wide contexts: the reference tables are
// direct-indexed and collision free, so a device table for 2^bits > budget
// must be an exact map.  Open addressing, key+1 stored (0 = empty), one
// writer (the owning lane).
__device__ __forceinline__ uint32_t *leaf_slot(const LeafParam &lp, uint8_t *lane_tbl, uint32_t ctx) {
    uint32_t *tbl = reinterpret_cast<uint32_t *>(lane_tbl + lp.tbl_off);
    if (!lp.use_hash) return tbl + ctx;
    uint32_t h = (ctx * 2654435761u) ^ (ctx >> 15);
    for (;;) {
        h &= lp.hash_mask;
        uint32_t k = tbl[2u * h];
        if (k == ctx + 1u) return tbl + 2u * h + 1u;
        if (k == 0u) { tbl[2u * h] = ctx + 1u; return tbl + 2u * h + 1u; }
        h++;
    }
}

// The leaf parameters are read with a loop-variant index.  hipcc strength-reduces such reads of the kernel
// arguments into s_load_dwordx2 from a byte-misaligned SGPR base (base = &leaf[l].<u8 field>), and the scalar
// memory unit drops the two low bits of the BASE, not of base+offset: leaf[l].tbl_off came back shifted for
// every l whose fields were not zero (found on MI355X as an aperture violation).  So the parameters are copied
// to LDS once with compile-time offsets and read from there.
__device__ __forceinline__ void stage_leaves(LeafParam *s_leaf, const GenericArgs &a) {
    if (threadIdx.x == 0) {
#pragma unroll
        for (int l = 0; l < W3_MAX_LEAVES; l++) s_leaf[l] = a.leaf[l];
    }
    __syncthreads();
}

template <bool DECODE>
__global__ void __launch_bounds__(64) k_generic(GenericArgs a) {
    __shared__ LeafParam s_leaf[W3_MAX_LEAVES];
    stage_leaves(s_leaf, a);
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= a.n_lanes) return;
    const uint32_t b = a.first_block + lane;
    const uint64_t off = (uint64_t)b * a.block_size;
    const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
    uint8_t *lane_tbl = a.tables + (uint64_t)lane * a.lane_stride;

    Encoder enc; Decoder dec;
    if (DECODE) dec.init(a.cin + a.coffs[b], a.clens[b]);
    else enc.init(a.stripes + (uint64_t)lane * a.stripe_cap, a.stripe_cap);

    uint64_t hist64 = 0; uint32_t t = 0;
    HuffState hs;
    for (uint32_t i = 0; i < len; i++) {
        uint32_t byte = DECODE ? 0u : a.in[off + i];
        for (int s = 7; s >= 0; s--) {
            // predict: leftmost leaf of maximal |p - 1/2| (BestOfTwo tree, models/mod.rs:67-69)
            uint32_t p = 32768u, best = 0u; bool first = true;
            uint32_t *slot[16];
            for (int l = 0; l < a.n_leaves; l++) {
                const LeafParam &lp = s_leaf[l];
                uint32_t pl = 32768u;
                slot[l] = nullptr;
                if (!lp.frozen) {                       // FrozenModel never adapts: Counter stays (0,0)
                    slot[l] = leaf_slot(lp, lane_tbl, leaf_ctx(lp, hist64, t, hs, a.huff));
                    pl = counter_p_packed(*slot[l]);
                }
                uint32_t d = opinion_dist(pl);
                if (first || d > best) { p = pl; best = d; first = false; }
            }
            uint32_t bit;
            if (DECODE) { bit = dec.decode(p); byte = (byte << 1) | bit; }
            else bit = (byte >> s) & 1u;
            // Model::update = adapt (train current ctx) then update (advance)  models/mod.rs:28-31
            for (int l = 0; l < a.n_leaves; l++)
                if (slot[l]) *slot[l] = counter_update_packed(*slot[l], bit);
            hist64 = (hist64 << 1) | bit;
            t++;
            if (!DECODE) enc.encode(bit, p);
        }
        if (a.n_huff) hs.push_byte(a.huff, a.n_huff, byte & 0xFFu);
        if (DECODE) a.dout[off + i] = (uint8_t)byte;
    }
    if (!DECODE) {
        if (a.out_bits) a.out_bits[b] = enc.stats_bits();
        uint32_t produced = enc.flush();
        a.out_len[b] = produced;
        if (produced > a.stripe_cap) atomicOr(a.overflow, 1u);
    }
}

// ---------------------------------------------------------------------------
// The same loop for 1..4 leaves with the leaf count a template parameter: the leaves' Counter loads of a step are
// independent, but the run-time leaf loop above waits for each before it starts the next (one memory round trip per
// leaf and step, ~1 us each for a lone lane).  Here all addresses are formed first and the loads issued together;
// an exact-map leaf issues its FIRST probe with them and only walks further (serially) on a collision.
// ---------------------------------------------------------------------------
template <bool DECODE, int NL>
__global__ void __launch_bounds__(64) k_generic_nl(GenericArgs a) {
    __shared__ LeafParam s_leaf[W3_MAX_LEAVES];
    stage_leaves(s_leaf, a);
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= a.n_lanes) return;
    const uint32_t b = a.first_block + lane;
    const uint64_t off = (uint64_t)b * a.block_size;
    const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
    uint8_t *lane_tbl = a.tables + (uint64_t)lane * a.lane_stride;
    const LeafParam *lp = s_leaf;   // (a register copy goes to scratch: ACHistory indexes lp.table dynamically)

    Encoder enc; Decoder dec;
    if (DECODE) dec.init(a.cin + a.coffs[b], a.clens[b]);
    else enc.init(a.stripes + (uint64_t)lane * a.stripe_cap, a.stripe_cap);

    uint64_t hist64 = 0; uint32_t t = 0;
    HuffState hs;
    for (uint32_t i = 0; i < len; i++) {
        uint32_t byte = DECODE ? 0u : a.in[off + i];
        for (int s = 7; s >= 0; s--) {
            uint32_t *slot[NL]; uint32_t val[NL], key[NL], ctx[NL];
            // 1. every leaf's address, then every leaf's load
#pragma unroll
            for (int l = 0; l < NL; l++) {
                slot[l] = nullptr; val[l] = 0u; key[l] = 0u; ctx[l] = 0u;
                if (!lp[l].frozen) {
                    ctx[l] = leaf_ctx(lp[l], hist64, t, hs, a.huff);
                    uint32_t *tbl = reinterpret_cast<uint32_t *>(lane_tbl + lp[l].tbl_off);
                    if (!lp[l].use_hash) slot[l] = tbl + ctx[l];
                    else slot[l] = tbl + 2u * (((ctx[l] * 2654435761u) ^ (ctx[l] >> 15)) & lp[l].hash_mask);   // first probe (leaf_slot)
                }
            }
#pragma unroll
            for (int l = 0; l < NL; l++) {
                if (slot[l]) {
                    if (!lp[l].use_hash) val[l] = *slot[l];
                    else { const uint2 kv = *reinterpret_cast<const uint2 *>(slot[l]); key[l] = kv.x; val[l] = kv.y; }
                }
            }
            // 2. exact-map leaves: hit, claim an empty slot, or keep probing (leaf_slot semantics)
#pragma unroll
            for (int l = 0; l < NL; l++) {
                if (slot[l] && lp[l].use_hash) {
                    if (key[l] == ctx[l] + 1u) slot[l] += 1;
                    else if (key[l] == 0u) { slot[l][0] = ctx[l] + 1u; slot[l] += 1; val[l] = 0u; }
                    else { slot[l] = leaf_slot(lp[l], lane_tbl, ctx[l]); val[l] = *slot[l]; }
                }
            }
            // 3. predict: leftmost leaf of maximal |p - 1/2| (BestOfTwo tree, models/mod.rs:67-69)
            uint32_t p = 32768u, best = 0u;
#pragma unroll
            for (int l = 0; l < NL; l++) {
                const uint32_t pl = slot[l] ? counter_p_packed(val[l]) : 32768u;
                const uint32_t d = opinion_dist(pl);
                if (l == 0 || d > best) { p = pl; best = d; }
            }
            uint32_t bit;
            if (DECODE) { bit = dec.decode(p); byte = (byte << 1) | bit; }
            else bit = (byte >> s) & 1u;
            // Model::update = adapt (train current ctx) then update (advance)  models/mod.rs:28-31
#pragma unroll
            for (int l = 0; l < NL; l++)
                if (slot[l]) *slot[l] = counter_update_packed(val[l], bit);
            hist64 = (hist64 << 1) | bit;
            t++;
            if (!DECODE) enc.encode(bit, p);
        }
        if (a.n_huff) hs.push_byte(a.huff, a.n_huff, byte & 0xFFu);
        if (DECODE) a.dout[off + i] = (uint8_t)byte;
    }
    if (!DECODE) {
        if (a.out_bits) a.out_bits[b] = enc.stats_bits();
        uint32_t produced = enc.flush();
        a.out_len[b] = produced;
        if (produced > a.stripe_cap) atomicOr(a.overflow, 1u);
    }
}

}  // namespace w3
