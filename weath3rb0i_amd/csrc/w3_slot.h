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
// kernel is the north star's literal design: ONE WAVEFRONT LANE PER BLOCK, the hash map in HBM (2^log_cells Cells
// per lane, one 128-byte line each), the 12-bit state table staged in LDS.  Per nibble a lane
//   * has the cell of THIS nibble staged in LDS ([8][64] x 16 B: byte-granular dynamic indexing without scratch; eight
//     16-byte loads issued two nibbles earlier: the hash of the next nibble's context is a function of input bytes
//     only) and issues the loads of the NEXT nibble's cell,
//   * matches the tag
//     (or evicts: the policy hashmap.rs:64-68 leaves as TODO), walks 4 states (2 dependent LDS reads per bit),
//   * writes the slot's 32-byte sector back (two 16-byte stores), and stages the prefetched cell of the next nibble in
//     its place (or keeps the LDS copy when the next nibble falls into the same cell).
// All slot leaves of a model run in ONE launch (blockIdx.y = leaf): 4 leaves x 239 wavefronts at enwik9 size put
// one wavefront on nearly every SIMD.  Algorithmic HBM bytes: 2 x (96 B read + 96 B written) per input byte
// and leaf; the device layout actually reads 128 B and writes 32 B per nibble.
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
    uint8_t *dummy;       // >= 1 KiB sink for predicated-off stores
    int n_leaves;
    uint32_t dbg_flags;   // timing experiments only (results wrong): 2 = every prefetch reads cell 0, 4 = no cell write-back
    SlotLeaf leaf[W3_MAX_SLOT_LEAVES];
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's uint4 class went to scratch)

// DEVICE LAYOUT of a Cell in HBM (this kernel's own; the reference's packed 96-byte layout, hashmap.rs:31-36, is kept
// byte-exact by k_cm): one 128-byte line = 4 sectors of 32 bytes, sector id = slot id = 15 states as u16 (node order of
// Slot::get_idx: node = (1 << bit_id) - 1 + nib_ctx, i.e. idx / 3) followed by the slot's 12-bit tag as u16.
// Same logical content (4 tags + 60 12-bit states, all zero in a fresh map); a nibble then reads ONE line and rewrites
// ONE 32-byte sector, with aligned u16 accesses instead of nibble arithmetic.
#define W3_CELL_STRIDE 128ull

// The lane's cell in LDS: chunk q (16 bytes) of lane l is s_cell[q][l]; u16 element x of the cell sits at
// u16 offset (x >> 3) * 512 + (x & 7) from the lane's base.
__device__ __forceinline__ uint32_t cx(uint32_t x) { return (x >> 3) * 512u + (x & 7u); }

// Everything one nibble does on the cell staged in LDS: slot lookup (or eviction), the four state steps.
// Returns the slot id; p[0..3] = StateTable::p of the four states BEFORE their update.
__device__ __forceinline__ uint32_t slot_nibble(lds_u16 *cb, const lds_u64 *st, uint32_t tag, uint32_t nib, uint32_t p[4]) {
    // Cell::get_slot (hashmap.rs:42-63): compare the tags of slot 3, 2, 1, 0 in this order
    const uint32_t t3 = cb[cx(63u)], t2 = cb[cx(47u)], t1 = cb[cx(31u)], t0 = cb[cx(15u)];
    int id = tag == t3 ? 3 : tag == t2 ? 2 : tag == t1 ? 1 : tag == t0 ? 0 : -1;
    if (id < 0) {
        // miss (hashmap.rs:64-68 TODO; policy as in w3_cm.h slot_select): victim = fewest observations in the
        // slot's first-bit state, candidates in the order 1, 0, 2, 3; tag stored, 15 states cleared
        const uint32_t f1 = cb[cx(16u)], f0 = cb[cx(0u)], f2 = cb[cx(32u)], f3 = cb[cx(48u)];
        uint32_t best = (uint32_t)(st[f1] >> 48); id = 1;
        uint32_t c = (uint32_t)(st[f0] >> 48); if (c < best) { best = c; id = 0; }
        c = (uint32_t)(st[f2] >> 48); if (c < best) { best = c; id = 2; }
        c = (uint32_t)(st[f3] >> 48); if (c < best) { best = c; id = 3; }
        W3_LDS_FENCE();
        typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
        lds_u32x4 *sec = (lds_u32x4 *)(cb + cx(16u * (uint32_t)id));
        u32x4 z; z.x = 0u; z.y = 0u; z.z = 0u; z.w = 0u;
        sec[0] = z;
        z.w = tag << 16;
        sec[64] = z;            // next chunk row: 64 lanes x 16 bytes further
        W3_LDS_FENCE();
    }
    // the nibble's four states (Slot::get_nib / set_nib, hashmap.rs:114-128), StateTable::p / next (state_table/mod.rs)
    uint32_t nib_ctx = 0u;
    const uint32_t base = 16u * (uint32_t)id;
#pragma unroll
    for (int bit_id = 0; bit_id < 4; bit_id++) {
        const uint32_t bit = (nib >> (3 - bit_id)) & 1u;
        const uint32_t x = cx(base + (1u << bit_id) - 1u + nib_ctx);   // Slot::get_idx / 3, hashmap.rs:80-84
        const uint32_t sv = cb[x];
        const uint64_t e = st[sv];
        p[bit_id] = (uint32_t)e & 0xFFFFu;
        cb[x] = (uint16_t)(bit ? ((uint32_t)(e >> 32) & 0xFFFFu) : ((uint32_t)e >> 16));
        W3_LDS_FENCE();
        nib_ctx = (nib_ctx << 1) | bit;
    }
    return (uint32_t)id;
}

__global__ void __launch_bounds__(64) k_slot(SlotArgs a) {
    __shared__ uint2 s_st[kStSize];
    __shared__ u32x4 s_cell[8][64];     // [chunk][owner lane]: the cell of every lane's current nibble
    __shared__ uint32_t s_x[3][64];     // owner -> holder mailboxes: [0] cell index to prefetch, [1] write-back word, [2] staging flag
    for (uint32_t i = threadIdx.x; i < (uint32_t)kStSize; i += 64u) s_st[i] = a.st[i];
    __syncthreads();
    typedef __attribute__((address_space(1))) u32x4 g_u32x4;
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    const uint32_t tid = threadIdx.x;
    const SlotLeaf &lf = a.leaf[blockIdx.y];
    const uint32_t lane0 = blockIdx.x * 64u, nl = a.n_lanes;
    // ---- owner role: lane `tid` owns block first_block + lane0 + tid (model state, hashes, state walks) ----
    const uint32_t olane = min(lane0 + tid, nl - 1u);                     // lanes past the batch mirror the last one, with len 0
    const uint64_t off = (uint64_t)(a.first_block + olane) * a.block_size;
    uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
    const uint32_t last = len - 1u;
    if (lane0 + tid >= nl) len = 0u;
    const uint8_t *blk = a.in + off;
    g_u32x4 *Pout = (g_u32x4 *)(lf.P + off);
    g_u32x4 *sink = (g_u32x4 *)a.dummy + tid;                             // where predicated-off stores go
    lds_u16 *cb = (lds_u16 *)&s_cell[0][tid];
    const lds_u64 *st = (const lds_u64 *)&s_st[0];
    const uint32_t order = lf.order, lshift = 64u - lf.log_cells;
    uint32_t maxlen = len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, d, 64));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    // ---- holder role: global memory is touched LINE-WISE.  In load kq, lane tid fetches 16-byte chunk (tid & 7) of the
    // cell of owner 8 kq + (tid >> 3): the 8 lanes of an owner cover its 128-byte line in one request (a lane loading its
    // own cell chunk by chunk issues 8 scattered requests, and the request count is what bounds this kernel, DESIGN.md
    // §2.5).  In store s, lanes 2o', 2o'+1 write the two chunks of the slot sector of owner 32 s + o'.
    const uint32_t hc = tid & 7u, ho = tid >> 3;
    uint8_t *tbl_leaf = a.tables + lf.tbl_off;
    uint8_t *gbase[8];
#pragma unroll
    for (int kq = 0; kq < 8; kq++) gbase[kq] = tbl_leaf + (uint64_t)min(lane0 + 8u * kq + ho, nl - 1u) * a.lane_stride + 16u * hc;
    uint8_t *sbase[2];
#pragma unroll
    for (int sq = 0; sq < 2; sq++) sbase[sq] = tbl_leaf + (uint64_t)min(lane0 + 32u * sq + (tid >> 1), nl - 1u) * a.lane_stride;
    lds_u32x4 *hcell = (lds_u32x4 *)&s_cell[hc][ho];                      // + 8 kq owners
    lds_u32 *mb0 = (lds_u32 *)&s_x[0][0], *mb1 = (lds_u32 *)&s_x[1][0], *mb2 = (lds_u32 *)&s_x[2][0];

    // Software pipeline over the nibbles ("events"; event e = nibble e & 1 of byte e >> 1).  The cell of event e is staged in
    // LDS; the cells of e+1 and e+2 are in flight / in the holders' registers (B[half of the event]); the loads of e+2 are
    // issued when e starts.  A prefetched cell is stale if an event processed after its loads were issued wrote the same
    // cell: e+1's cell equal to e's (keep the LDS copy) or to e-1's (take G, e-1's cell as it was written back).
    // Every global load and store is unconditional and their number per event is fixed (predicated-off ones go to a sink):
    // hipcc waits vmcnt(0) — i.e. for the prefetches just issued — as soon as a branch makes the count uncertain.
    uint32_t b0 = blk[0], b1 = blk[min(1u, last)], b2 = blk[min(2u, last)];
    uint64_t hist = 0ull;
    uint64_t h_cur = slot_hash(order, 0ull, false, 0u), h_1 = slot_hash(order, 0ull, true, b0 >> 4);
    uint32_t c_cur = (uint32_t)(h_cur >> lshift), c_1 = (uint32_t)(h_1 >> lshift), c_prev = 0xFFFFFFFFu;
    u32x4 B0[8], B1[8], G0[8], G1[8];
    mb0[tid] = c_cur; mb1[tid] = c_1;
    W3_LDS_FENCE();
#pragma unroll
    for (int kq = 0; kq < 8; kq++) {
        B0[kq] = *(const g_u32x4 *)(gbase[kq] + (uint64_t)mb0[8u * kq + ho] * W3_CELL_STRIDE);
        B1[kq] = *(const g_u32x4 *)(gbase[kq] + (uint64_t)mb1[8u * kq + ho] * W3_CELL_STRIDE);
        G0[kq] = 0u; G1[kq] = 0u;
    }
#pragma unroll
    for (int kq = 0; kq < 8; kq++) hcell[8 * kq] = B0[kq];
    W3_LDS_FENCE();
    for (uint32_t i = 0; i < maxlen; i++) {
        const uint32_t byte = b0;
        const uint32_t b3 = blk[min(i + 3u, last)];
        const bool act = i < len, act_next = i + 1u < len;
        uint32_t pw[4];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            u32x4 (&Bn)[8] = half == 0 ? B1 : B0;    // the cells of e+1 (loaded while e-1 ran)
            u32x4 (&Bl)[8] = half == 0 ? B0 : B1;    // receive the cells of e+2
            u32x4 (&Gp)[8] = half == 0 ? G1 : G0;    // e-1's cells as written back
            u32x4 (&Gc)[8] = half == 0 ? G0 : G1;    // receive e's
            // owner: where is the cell of e+2?   holders: fetch it
            const uint64_t h_2 = half == 0 ? slot_hash(order, (hist << 8) | byte, false, 0u)
                                           : slot_hash(order, (hist << 8) | byte, true, b1 >> 4);
            const uint32_t c_2 = (uint32_t)(h_2 >> lshift);
            mb0[tid] = (a.dbg_flags & 2u) ? 0u : c_2;
            W3_LDS_FENCE();
#pragma unroll
            for (int kq = 0; kq < 8; kq++) Bl[kq] = *(const g_u32x4 *)(gbase[kq] + (uint64_t)mb0[8u * kq + ho] * W3_CELL_STRIDE);
            // owner: this nibble on the staged cell
            uint32_t pq[4] = {0u, 0u, 0u, 0u}, id = 0u;
            if (act) id = slot_nibble(cb, st, (uint32_t)h_cur & 0xFFFu, half == 0 ? (byte >> 4) : (byte & 15u), pq);
            pw[2 * half] = pq[0] | (pq[1] << 16);
            pw[2 * half + 1] = pq[2] | (pq[3] << 16);
            mb1[tid] = c_cur | (id << 24) | ((act && !(a.dbg_flags & 4u)) ? (1u << 26) : 0u);
            W3_LDS_FENCE();
            // holders: write the slot's 32-byte sector back (two lanes per owner), keep every cell as written (G)
#pragma unroll
            for (int sq = 0; sq < 2; sq++) {
                const uint32_t o = 32u * sq + (tid >> 1), wb = mb1[o];
                const uint32_t chunk = 2u * ((wb >> 24) & 3u) + (tid & 1u);
                const u32x4 data = s_cell[chunk][o];
                g_u32x4 *dst = (wb >> 26) ? (g_u32x4 *)(sbase[sq] + (uint64_t)(wb & 0xFFFFFFu) * W3_CELL_STRIDE + 16u * chunk) : sink;
                *dst = data;
            }
            // (G is only ever read when some lane's event e+2 returns to the cell of e: skip the read-back otherwise)
            if (__ballot(c_2 == c_cur)) {
#pragma unroll
                for (int kq = 0; kq < 8; kq++) Gc[kq] = hcell[8 * kq];
            }
            // owner: is the prefetched cell of e+1 still good?   holders: stage it
            const bool nact = half == 0 ? act : act_next;
            mb2[tid] = (!nact || c_1 == c_cur) ? 0u : (c_1 == c_prev ? 1u : 2u);
            W3_LDS_FENCE();
#pragma unroll
            for (int kq = 0; kq < 8; kq++) {
                const uint32_t f = mb2[8u * kq + ho];
                const u32x4 v = f == 1u ? Gp[kq] : Bn[kq];
                if (f) hcell[8 * kq] = v;
            }
            W3_LDS_FENCE();
            c_prev = c_cur; h_cur = h_1; c_cur = c_1; h_1 = h_2; c_1 = c_2;
        }
        hist = (hist << 8) | byte;
        b0 = b1; b1 = b2; b2 = b3;
        {
            u32x4 pv; pv.x = pw[0]; pv.y = pw[1]; pv.z = pw[2]; pv.w = pw[3];
            g_u32x4 *dst = act ? Pout + i : sink;
            *dst = pv;
        }
    }
}

}  // namespace w3
