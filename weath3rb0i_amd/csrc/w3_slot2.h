// w3_slot2.h — slot-state leaves WITHOUT a hash map in HBM: the nibble events of a block sorted by Cell and replayed with the
// open Cell in LDS.  The form for inputs of FEW blocks (BASELINE configs[2] at its literal enwik8 size, configs[4]'s 256 KiB
// blocks): k_slot (w3_slot.h) is one lane per block walking a table in HBM, so a call costs at least the lone-lane latency of a
// whole block (about 200 ms per 64 KiB of block) however few blocks there are; here the work of a block is spread over a
// wavefront and costs what its bytes cost.  At enwik9 size k_slot's 15,259 lanes hide that latency and it is the faster one
// (375 against 460 ms: profiles/r3_slot_sorted_experiment/), so twophase_predict_b picks by block count.
//
// A slot-state leaf (BUILD-DEFINED model over the reference's primitives: hashmap.rs Cell/Slot, state_table/naive.rs; DESIGN.md
// section 2.4) looks up, once per nibble ("event"), the Cell of hash(order, previous bytes, nibble marker).  The encoder knows
// every event's Cell in advance (the hash is a function of input bytes only), and a Cell's history is only ever touched by ITS
// events.  So:
//   1. k_slot_events   one 8-byte record per event:  cell << 48 | tag << 36 | nibble << 32 | event index  — everything the replay
//                      needs, so that it never gathers input windows — and the digit histograms of the two sort passes
//   2. k_slot_sort     stable LSD partition of a block's records by cell, 8 bits per pass, through LDS tiles (the machinery of
//                      k_partition8: a record's slot in the tile is one returning LDS add on its bin's cursor — lane-ordered)
//   3. k_slot_replay   one WAVEFRONT per (block, leaf); a lane replays the events of a range of Cells (the sort's last-pass
//                      bins: contiguous in the sorted array) one by one: the open Cell lives in LDS (128 B per lane), starts empty
//                      when the cell index changes and is simply dropped when its last event is through — nothing of it ever goes
//                      to memory.  Tag match / eviction / four state steps as in k_slot (hashmap.rs:42-71, 80-128), the four
//                      state reads of a nibble issued together (their addresses depend on the nibble's bits, not on the states
//                      read).  The replay also CHECKS the sort it relies on (cells ascending inside a bin, events ascending
//                      inside a cell): a violation — returning LDS adds not resolved in lane order — is counted in the call's
//                      flag word 2, which makes the host code the call again on the ballot path, where k_slot runs.
// HBM traffic per event: 8 B record written, 1 or 2 passes of (8 r + 8 w), 8 B read, 8 B of probabilities written (scattered
// into the block's own stream) = 56 B against k_slot's 160 B; no table memory, no zero-fill.
#pragma once
#include "w3_predict.h"
#include "w3_predict_wave.h"
#include "w3_slot.h"

namespace w3 {

struct Slot2Args {
    const uint8_t *in;
    uint64_t n;
    uint32_t block_size, nblocks;
    uint64_t *keys_a, *keys_b;   // [n_leaves][2 n] each
    uint32_t *hist;       // [n_leaves][nblocks][512]: counts of the low 8 cell bits, of the bits above
    const uint2 *st;      // [kStSize] {prob | next0 << 16, next1 | conf << 16}
    uint32_t *fault;      // the call's flag word 2 (order violations seen by the replay), or null
    uint8_t *dummy;       // >= 1 KiB sink for predicated-off stores
    int n_leaves;
    SlotLeaf leaf[W3_MAX_SLOT_LEAVES];   // (tbl_off unused)
};

#define W3_S2_CELL_SH 48u
#define W3_S2_TAG_SH 36u
#define W3_S2_NIB_SH 32u

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_slot_events(Slot2Args a) {
    __shared__ uint32_t s_hist[512];
    const uint32_t lane = threadIdx.x & 63u;
    const SlotLeaf &lf = a.leaf[blockIdx.y];
    const uint32_t order = lf.order, lshift = 64u - lf.log_cells;
    uint64_t *keys = a.keys_a + (uint64_t)blockIdx.y * 2u * a.n;
    for (uint32_t b = blockIdx.x; b < a.nblocks; b += gridDim.x) {
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
        const uint8_t *blk = a.in + off;
        const bool first = off == 0;
#pragma unroll
        for (int k = 0; k < 8; k++) s_hist[k * 64 + lane] = 0u;
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const uint32_t nev = 2u * len;
        for (uint32_t e0 = 0; e0 < nev; e0 += 64u) {
            const uint32_t e = e0 + lane;
            const bool valid = e < nev;
            const uint32_t i = min(e, nev - 1u) >> 1, half = e & 1u;
            const uint64_t W = wave_window(blk, i, first);
            const uint32_t byte = (uint32_t)W & 0xFFu;
            const uint64_t h = slot_hash(order, W >> 8, half != 0u, byte >> 4);
            const uint32_t cell = (uint32_t)(h >> lshift);
            const uint32_t nib = half ? (byte & 15u) : (byte >> 4);
            if (valid) {
                keys[2u * off + e] = ((uint64_t)cell << W3_S2_CELL_SH) | ((uint64_t)((uint32_t)h & 0xFFFu) << W3_S2_TAG_SH) | ((uint64_t)nib << W3_S2_NIB_SH) | e;
                __hip_atomic_fetch_add(&s_hist[cell & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&s_hist[256u + (cell >> 8)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        uint32_t *gh = a.hist + ((uint64_t)blockIdx.y * a.nblocks + b) * 512u;
#pragma unroll
        for (int k = 0; k < 8; k++) gh[k * 64 + lane] = s_hist[k * 64 + lane];
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------
#define W3_S2_TILE 2048u
#define W3_S2_ROUNDS (W3_S2_TILE / 64u)

// PASS 0: digit = cell bits 0..7, keys_a -> keys_b;  PASS 1: digit = cell bits 8..15, keys_b -> keys_a.
template <int PASS>
__global__ void __launch_bounds__(64) k_slot_sort(Slot2Args a) {
    __shared__ uint64_t tile[W3_S2_TILE];
    __shared__ uint32_t gcur[256], tcnt[256], tstart[256], tcur[256];
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t sh = W3_S2_CELL_SH + 8u * PASS;
    const uint64_t *src_l = (PASS == 0 ? a.keys_a : a.keys_b) + (uint64_t)blockIdx.y * 2u * a.n;
    uint64_t *dst_l = (PASS == 0 ? a.keys_b : a.keys_a) + (uint64_t)blockIdx.y * 2u * a.n;
    for (uint32_t b = blockIdx.x; b < a.nblocks; b += gridDim.x) {
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
        const uint32_t nev = 2u * len, last = nev - 1u;
        const uint64_t *src = src_l + 2u * off;
        uint64_t *out = dst_l + 2u * off;
        const uint32_t *gh = a.hist + ((uint64_t)blockIdx.y * a.nblocks + b) * 512u + 256u * PASS;
#pragma unroll
        for (int k = 0; k < 4; k++) tcnt[k * 64 + lane] = gh[k * 64 + lane];
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        wave_excl_scan_256(tcnt, gcur, nullptr);
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t t0 = 0; t0 < nev; t0 += W3_S2_TILE) {
            const uint32_t tlen = min(W3_S2_TILE, nev - t0);
            uint64_t key[W3_S2_ROUNDS];
#pragma unroll
            for (uint32_t r = 0; r < W3_S2_ROUNDS; r++) key[r] = src[min(t0 + r * 64u + lane, last)];
#pragma unroll
            for (int k = 0; k < 4; k++) tcnt[k * 64 + lane] = 0u;
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t r = 0; r < W3_S2_ROUNDS; r++)
                if (r * 64u + lane < tlen) __hip_atomic_fetch_add(&tcnt[(uint32_t)(key[r] >> sh) & 0xFFu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            wave_excl_scan_256(tcnt, tstart, tcur);
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // stable scatter into the tile: rounds in time order, lanes in order inside the returning add (atomic_round's property)
#pragma unroll
            for (uint32_t r = 0; r < W3_S2_ROUNDS; r++) {
                if (r * 64u + lane < tlen) {
                    const uint32_t slot = __hip_atomic_fetch_add(&tcur[(uint32_t)(key[r] >> sh) & 0xFFu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    tile[slot] = key[r];
                }
            }
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // copy out: element k of the sorted tile belongs to bin d at run offset k - tstart[d]
#pragma unroll
            for (uint32_t r = 0; r < W3_S2_ROUNDS; r++) {
                const uint32_t k = r * 64u + lane;
                if (k < tlen) {
                    const uint64_t kv = tile[k];
                    const uint32_t d = (uint32_t)(kv >> sh) & 0xFFu;
                    out[gcur[d] + (k - tstart[d])] = kv;
                }
            }
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++) gcur[k * 64 + lane] += tcnt[k * 64 + lane];
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------
// One nibble on the Cell staged in LDS: slot lookup (or eviction), the four state steps with their reads issued together.
__device__ __forceinline__ void slot_nibble2(lds_u16 *cb, const lds_u64 *st, uint32_t tag, uint32_t nib, uint32_t p[4]) {
    // Cell::get_slot (hashmap.rs:42-63): compare the tags of slot 3, 2, 1, 0 in this order
    const uint32_t t3 = cb[cx(63u)], t2 = cb[cx(47u)], t1 = cb[cx(31u)], t0 = cb[cx(15u)];
    int id = tag == t3 ? 3 : tag == t2 ? 2 : tag == t1 ? 1 : tag == t0 ? 0 : -1;
    if (id < 0) {
        // miss (hashmap.rs:64-68 TODO; policy of w3_cm.h slot_select): victim = fewest observations in the slot's first-bit state,
        // candidates in the order 1, 0, 2, 3; tag stored, 15 states cleared
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
    // Slot::get_nib / set_nib (hashmap.rs:114-128): node k of the path = (1 << k) - 1 + (the nibble's top k bits); the four addresses
    // are known before any state is read, so the reads go out together, then the four table rows, then the four writes
    const uint32_t base = 16u * (uint32_t)id;
    uint32_t x[4], sv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = cx(base + (1u << k) - 1u + (nib >> (4 - k)));
#pragma unroll
    for (int k = 0; k < 4; k++) sv[k] = cb[x[k]];
    uint64_t e[4];
#pragma unroll
    for (int k = 0; k < 4; k++) e[k] = st[sv[k]];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t bit = (nib >> (3 - k)) & 1u;
        p[k] = (uint32_t)e[k] & 0xFFFFu;
        cb[x[k]] = (uint16_t)(bit ? ((uint32_t)(e[k] >> 32) & 0xFFFFu) : ((uint32_t)e[k] >> 16));
    }
    W3_LDS_FENCE();
}

#define W3_S2_WAVES 4   // wavefronts per workgroup of k_slot_replay (they share the state table in LDS)
#define W3_S2_BATCH 8u  // records per lane and memory wait

// One WAVEFRONT per (block, leaf); lane l replays the events of the sort's last-pass bins l, l + 64, l + 128, l + 192 — cell
// ranges whose events are contiguous in the sorted array (start = exclusive scan of the block's digit counts) — one after the
// other, serially, with the open Cell in LDS.  All 64 lanes work on ONE block: the 8-byte probability stores land in the
// block's own stream.
__global__ void __launch_bounds__(64 * W3_S2_WAVES) k_slot_replay(Slot2Args a, int two_passes) {
    __shared__ uint2 s_st[kStSize];
    __shared__ u32x4 s_cell[W3_S2_WAVES][8][64];   // [wave][chunk][lane]: the open Cell of every lane
    __shared__ uint32_t s_cnt[W3_S2_WAVES][256], s_start[W3_S2_WAVES][256];
    for (uint32_t i = threadIdx.x; i < (uint32_t)kStSize; i += 64u * W3_S2_WAVES) s_st[i] = a.st[i];
    __syncthreads();
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) u32x2 g_uint2;
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const SlotLeaf &lf = a.leaf[blockIdx.y];
    const uint32_t b = blockIdx.x * W3_S2_WAVES + wv;
    if (b >= a.nblocks) return;   // (no barrier below)
    const uint64_t off = (uint64_t)b * a.block_size;
    const uint64_t *keys = (two_passes ? a.keys_a : a.keys_b) + (uint64_t)blockIdx.y * 2u * a.n + 2u * off;
    // the bins of the last pass that moved anything (a leaf of at most 2^8 Cells goes through the second pass unchanged: one bin)
    const uint32_t *gh = a.hist + ((uint64_t)blockIdx.y * a.nblocks + b) * 512u + ((two_passes && lf.log_cells > 8u) ? 256u : 0u);
    g_uint2 *Pout = (g_uint2 *)(lf.P + off);
    lds_u16 *cb = (lds_u16 *)&s_cell[wv][0][lane];
    lds_u32x4 *cq = (lds_u32x4 *)&s_cell[wv][0][lane];
    const lds_u64 *st = (const lds_u64 *)&s_st[0];
#pragma unroll
    for (int k = 0; k < 4; k++) s_cnt[wv][k * 64 + lane] = gh[k * 64 + lane];
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    wave_excl_scan_256(s_cnt[wv], s_start[wv], nullptr);
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    uint32_t bs_[4], bc_[4], tot = 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) { bs_[q] = s_start[wv][q * 64 + lane]; bc_[q] = s_cnt[wv][q * 64 + lane]; tot += bc_[q]; }
    uint32_t maxtot = tot;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxtot = max(maxtot, (uint32_t)__shfl_xor((int)maxtot, d, 64));
    maxtot = __builtin_amdgcn_readfirstlane(maxtot);
    // this lane's events as one sequence: bins q = 0..3 one after the other; event k of the lane sits at idx(k)
    const uint32_t c0 = bc_[0], c1 = c0 + bc_[1], c2 = c1 + bc_[2];
    auto idx = [&](uint32_t k) -> uint32_t {
        const uint32_t kk = min(k, tot ? tot - 1u : 0u);   // (clamped: every load is unconditional and in range)
        return kk < c0 ? bs_[0] + kk : kk < c1 ? bs_[1] + (kk - c0) : kk < c2 ? bs_[2] + (kk - c1) : bs_[3] + (kk - c2);
    };
    g_uint2 *sink = (g_uint2 *)(a.dummy + 8u * lane);
    uint32_t cur_cell = 0xFFFFFFFFu, prev_e = 0u, bad = 0u;
    // Records travel in batches of W3_S2_BATCH per lane: the loads of batch k + 1 are issued before batch k is replayed and every
    // store is unconditional (idle lanes write to the sink), so the loop waits for memory once per batch, not once per event (gfx9
    // counts loads and stores in one counter: waiting for a record also waits for the scattered store issued just before it).
    uint64_t kn[W3_S2_BATCH];
#pragma unroll
    for (uint32_t j = 0; j < W3_S2_BATCH; j++) kn[j] = keys[idx(j)];
    for (uint32_t k0 = 0; k0 < maxtot; k0 += W3_S2_BATCH) {
        uint64_t kc[W3_S2_BATCH];
#pragma unroll
        for (uint32_t j = 0; j < W3_S2_BATCH; j++) kc[j] = kn[j];
#pragma unroll
        for (uint32_t j = 0; j < W3_S2_BATCH; j++) kn[j] = keys[idx(k0 + W3_S2_BATCH + j)];
#pragma unroll
        for (uint32_t j = 0; j < W3_S2_BATCH; j++) {
            const bool act = k0 + j < tot;
            const uint64_t key = kc[j];
            const uint32_t cell = (uint32_t)(key >> W3_S2_CELL_SH), e = (uint32_t)key;
            const uint32_t tag = (uint32_t)(key >> W3_S2_TAG_SH) & 0xFFFu, nib = (uint32_t)(key >> W3_S2_NIB_SH) & 15u;
            uint32_t pq[4] = {0u, 0u, 0u, 0u};
            if (act) {
                if (cell != cur_cell) {   // the previous Cell's events are through: this one starts empty (a fresh HashMap is all zeros)
                    // the sort this replay relies on: inside a lane's bins the cells ascend (bins l, l + 64, .. are ascending ranges)
                    bad |= (cur_cell != 0xFFFFFFFFu && cell < cur_cell) ? 1u : 0u;
                    u32x4 z; z.x = 0u; z.y = 0u; z.z = 0u; z.w = 0u;
#pragma unroll
                    for (int qq = 0; qq < 8; qq++) cq[64 * qq] = z;
                    W3_LDS_FENCE();
                    cur_cell = cell;
                } else bad |= e <= prev_e ? 1u : 0u;   // ... and inside a cell the events keep their time order (the partition is stable)
                prev_e = e;
                slot_nibble2(cb, st, tag, nib, pq);
            }
            u32x2 v; v.x = pq[0] | (pq[1] << 16); v.y = pq[2] | (pq[3] << 16);
            g_uint2 *dst = act ? Pout + e : sink;
            *dst = v;
        }
    }
    if (bad && a.fault) atomicAdd(a.fault, 1u);
}

}  // namespace w3
