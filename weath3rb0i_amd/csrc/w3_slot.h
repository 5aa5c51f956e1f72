// w3_slot.h — PREDICT kernel of the slot-state leaves (two-phase encoder, gfx950).
//
// A slot-state leaf (BUILD-DEFINED model over the reference's primitives)
// looks up, once per nibble, the Cell of hash(order, previous bytes, nibble marker) in a per-block HashMap
// (hashmap.rs:1-71: cell by the HIGH hash bits, four 12-bit tags, 4 slots x 15 12-bit states = 96 bytes) and walks
// the 4 NaiveStateTable states on the nibble's path (hashmap.rs:114-128, state_table/naive.rs).
//
// Like a Counter leaf, its prediction never depends on the other leaves, the mixer or the coder, and the encoder
// knows every future context — so the leaf runs in the predict phase and leaves a u16 stream.  What is serial is
// the table itself (hits, evictions and state walks are a function of the whole history of the block), so this
// kernel is the north star's literal design: ONE WAVEFRONT LANE PER BLOCK, the hash map in HBM (2^log_cells x 96 B
// per lane), the 12-bit state table staged in LDS.  Per nibble a lane
//   * has the cell of THIS nibble staged in LDS ([6][64] x 16 B: byte-granular dynamic indexing without scratch; six
//     16-byte loads issued one nibble earlier: the hash of the next nibble's context is a function of input bytes
//     only) and issues the loads of the NEXT nibble's cell,
//   * matches the tag
//     (or evicts: the policy hashmap.rs:64-68 leaves as TODO), walks 4 states (2 dependent LDS reads per bit),
//   * writes the cell back with six 16-byte stores, and stages the prefetched cell of the next nibble in its place
//     (or simply keeps the LDS copy when the next nibble falls into the same cell).
// All slot leaves of a model run in ONE launch (blockIdx.y = leaf): 4 leaves x 239 wavefronts at enwik9 size put
// one wavefront on nearly every SIMD, and the kernel is bound by HBM: 2 x (96 B read + 96 B written) per input byte
// and leaf — the algorithmic figure of SURVEY §8(d).
#pragma once
#include "w3_apm.h"
#include "w3_cm.h"

namespace w3 {

#define W3_MAX_SLOT_LEAVES 8

struct SlotLeaf {
    uint32_t order, log_cells;
    uint64_t tbl_off;     // byte offset of this leaf's HashMap inside the lane's table region (multiple of 16)
    uint4 *P;             // this leaf's stream (8 x u16 per input byte)
};

struct SlotArgs {
    const uint8_t *in;
    uint64_t n;
    uint32_t block_size, first_block, n_lanes;
    uint8_t *tables;      // lane l owns [tables + l * lane_stride, + lane_stride), zeroed before the launch
    uint64_t lane_stride;
    const uint2 *st;      // [kStSize] {prob | next0 << 16, next1 | conf << 16}
    int n_leaves;
    SlotLeaf leaf[W3_MAX_SLOT_LEAVES];
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's uint4 class went to scratch)

// The lane's cell in LDS: dword d of lane l is s_cell[d >> 2][l].{x,y,z,w}[d & 3]; `cb` points at s_cell[0][l].
__device__ __forceinline__ uint32_t cb_off(uint32_t d) { return (d >> 2) * 256u + (d & 3u); }

struct CellRef {          // the big-endian u16 at a byte offset of the staged cell (hashmap.rs:86-97 reads exactly this)
    uint32_t olo, ohi, lo, hi, sh;
};
__device__ __forceinline__ uint32_t cell_get16(lds_u32 *cb, uint32_t byte, CellRef &r) {
    const uint32_t d = byte >> 2;
    r.olo = cb_off(d);
    r.ohi = cb_off(d == 23u ? 22u : d + 1u);   // byte <= 94: the pair only spills out of dword 23 never; 22 is a harmless stand-in
    r.lo = cb[r.olo]; r.hi = cb[r.ohi];
    r.sh = 8u * (byte & 3u);
    const uint32_t q = (uint32_t)((((uint64_t)r.hi << 32) | r.lo) >> r.sh);
    return ((q & 0xFFu) << 8) | ((q >> 8) & 0xFFu);
}
__device__ __forceinline__ void cell_put16(lds_u32 *cb, const CellRef &r, uint32_t v) {
    const uint64_t x = ((v >> 8) & 0xFFu) | ((v & 0xFFu) << 8);
    const uint64_t Q = (((((uint64_t)r.hi << 32) | r.lo)) & ~(0xFFFFull << r.sh)) | (x << r.sh);
    cb[r.olo] = (uint32_t)Q;
    cb[r.ohi] = (uint32_t)(Q >> 32);   // unchanged unless the pair straddles two dwords
}

// zero the nibbles [s, s + 45) of the staged cell (the 15 states of one slot), s = 12 + 45 * id
template <int ID>
__device__ __forceinline__ void cell_clear_slot(lds_u32 *cb) {
    constexpr int s = 12 + 45 * ID, e = s + 45;
#pragma unroll
    for (int d = s / 8; d <= (e - 1) / 8; d++) {
        uint32_t keep = 0u;   // mask of the bits that survive (little-endian dword: byte b at bits 8b; even nibble = high half)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int nib = 8 * d + k;
            const bool in = nib >= s && nib < e;
            const int byte_in_dw = k >> 1, high = (k & 1) == 0;
            if (!in) keep |= (high ? 0xF0u : 0x0Fu) << (8 * byte_in_dw);
        }
        if (keep == 0u) cb[cb_off(d)] = 0u;
        else cb[cb_off(d)] &= keep;
    }
}

__global__ void __launch_bounds__(64) k_slot(SlotArgs a) {
    __shared__ uint2 s_st[kStSize];
    __shared__ u32x4 s_cell[6][64];
    __shared__ SlotLeaf s_leaf[W3_MAX_SLOT_LEAVES];
    for (uint32_t i = threadIdx.x; i < (uint32_t)kStSize; i += 64u) s_st[i] = a.st[i];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int l = 0; l < W3_MAX_SLOT_LEAVES; l++) s_leaf[l] = a.leaf[l];
    }
    __syncthreads();
    const SlotLeaf lf = s_leaf[blockIdx.y];
    const uint32_t lane = blockIdx.x * 64u + threadIdx.x;
    if (lane >= a.n_lanes) return;
    const uint32_t b = a.first_block + lane;
    const uint64_t off = (uint64_t)b * a.block_size;
    const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
    const uint8_t *blk = a.in + off;
    uint8_t *cells = a.tables + (uint64_t)lane * a.lane_stride + lf.tbl_off;
    uint4 *Pout = lf.P + off;
    lds_u32 *cb = (lds_u32 *)&s_cell[0][threadIdx.x];
    const lds_u64 *st = (const lds_u64 *)&s_st[0];   // {prob | next0 << 16, next1 | conf << 16} as one 64-bit LDS read
    const uint32_t order = lf.order, lshift = 64u - lf.log_cells;

    uint64_t hist = 0ull;
    uint64_t h = slot_hash(order, 0ull, false, 0u);
    uint64_t cidx = h >> lshift;
    {   // invariant of the loop below: the lane's LDS staging area holds the cell of the current nibble
        const u32x4 *cp = reinterpret_cast<const u32x4 *>(cells + cidx * 96ull);
        u32x4 cur[6];
#pragma unroll
        for (int q = 0; q < 6; q++) cur[q] = cp[q];
#pragma unroll
        for (int q = 0; q < 6; q++) s_cell[q][threadIdx.x] = cur[q];
        W3_LDS_FENCE();
    }
    const uint32_t last = len - 1u;
    uint32_t nbyte = blk[0];
    for (uint32_t i = 0; i < len; i++) {
        const uint32_t byte = nbyte;
        nbyte = blk[min(i + 1u, last)];
        uint32_t pw[4];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            // the next nibble's cell: loads in flight while this nibble is processed
            const uint64_t hn = half == 0 ? slot_hash(order, hist, true, byte >> 4) : slot_hash(order, (hist << 8) | byte, false, 0u);
            const uint64_t cidx_n = hn >> lshift;
            u32x4 nxt[6];
            {
                const u32x4 *cp = reinterpret_cast<const u32x4 *>(cells + cidx_n * 96ull);
#pragma unroll
                for (int q = 0; q < 6; q++) nxt[q] = cp[q];
            }
            W3_LDS_FENCE();
            // Cell::get_slot (hashmap.rs:42-63): the four 12-bit tags are the big-endian bytes 0..5; compare id 3, 2, 1, 0
            const uint32_t tag = (uint32_t)h & 0xFFFu;
            const uint32_t tw0 = cb[cb_off(0)], tw1 = cb[cb_off(1)];
            const uint64_t hc = ((uint64_t)__builtin_bswap32(tw0) << 16) | (__builtin_bswap32(tw1) >> 16);
            int id = -1;
            if (tag == (uint32_t)(hc & 0xFFFu)) id = 3;
            else if (tag == (uint32_t)((hc >> 12) & 0xFFFu)) id = 2;
            else if (tag == (uint32_t)((hc >> 24) & 0xFFFu)) id = 1;
            else if (tag == (uint32_t)((hc >> 36) & 0xFFFu)) id = 0;
            if (id < 0) {
                // miss (hashmap.rs:64-68 TODO; policy as in w3_cm.h slot_select): victim = fewest observations in the
                // slot's first-bit state, candidates in the order 1, 0, 2, 3; tag stored, 15 states cleared
                uint32_t best = 0xFFFFFFFFu;
                const int cand[4] = {1, 0, 2, 3};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t idx = 45u * (uint32_t)cand[k];
                    CellRef r;
                    const uint32_t v = cell_get16(cb, 6u + (idx >> 1), r);
                    const uint32_t s0 = (idx & 1u) ? (v & 0xFFFu) : (v >> 4);
                    const uint32_t conf = (uint32_t)(st[s0] >> 48);
                    if (conf < best) { best = conf; id = cand[k]; }
                }
                const uint32_t shv = 12u * (3u - (uint32_t)id);
                const uint64_t hc2 = (hc & ~(0xFFFull << shv)) | ((uint64_t)tag << shv);
                // bytes 0..5 back (big-endian), bytes 6, 7 of dword 1 kept
                W3_LDS_FENCE();
                cb[cb_off(0)] = __builtin_bswap32((uint32_t)(hc2 >> 16));
                cb[cb_off(1)] = (tw1 & 0xFFFF0000u) | (__builtin_bswap32((uint32_t)(hc2 << 16)) & 0x0000FFFFu);
                W3_LDS_FENCE();
                switch (id) {
                case 0: cell_clear_slot<0>(cb); break;
                case 1: cell_clear_slot<1>(cb); break;
                case 2: cell_clear_slot<2>(cb); break;
                default: cell_clear_slot<3>(cb); break;
                }
                W3_LDS_FENCE();
            }
            // the nibble's four states (Slot::get_nib / set_nib, hashmap.rs:114-128), StateTable::p / next (state_table/mod.rs)
            const uint32_t nib = half == 0 ? (byte >> 4) : (byte & 15u);
            uint32_t nib_ctx = 0u, pq[4];
#pragma unroll
            for (int bit_id = 0; bit_id < 4; bit_id++) {
                const uint32_t bit = (nib >> (3 - bit_id)) & 1u;
                const uint32_t idx = (3u << bit_id) + 3u * nib_ctx + 45u * (uint32_t)id - 3u;   // Slot::get_idx, hashmap.rs:80-84
                CellRef r;
                const uint32_t v = cell_get16(cb, 6u + (idx >> 1), r);
                const uint32_t sv = (idx & 1u) ? (v & 0xFFFu) : (v >> 4);
                const uint64_t e = st[sv];
                pq[bit_id] = (uint32_t)e & 0xFFFFu;
                const uint32_t ns = bit ? ((uint32_t)(e >> 32) & 0xFFFFu) : ((uint32_t)e >> 16);
                cell_put16(cb, r, (idx & 1u) ? ((v & 0xF000u) | ns) : ((ns << 4) | (v & 0xFu)));
                W3_LDS_FENCE();
                nib_ctx = (nib_ctx << 1) | bit;
            }
            pw[2 * half] = pq[0] | (pq[1] << 16);
            pw[2 * half + 1] = pq[2] | (pq[3] << 16);
            // write the cell back; the next nibble may be in the same cell: forward it from LDS
            u32x4 g[6];
#pragma unroll
            for (int q = 0; q < 6; q++) g[q] = s_cell[q][threadIdx.x];
            {
                u32x4 *cp = reinterpret_cast<u32x4 *>(cells + cidx * 96ull);
#pragma unroll
                for (int q = 0; q < 6; q++) cp[q] = g[q];
            }
            if (cidx_n != cidx) {
#pragma unroll
                for (int q = 0; q < 6; q++) s_cell[q][threadIdx.x] = nxt[q];
            }
            W3_LDS_FENCE();
            h = hn; cidx = cidx_n;
        }
        hist = (hist << 8) | byte;
        Pout[i] = make_uint4(pw[0], pw[1], pw[2], pw[3]);
    }
}

}  // namespace w3
