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
//                      to memory.  Tag match / eviction / four state steps as in k_slot (hashmap.rs:42-71, 80-128) with the slots' tags and
//                      first-bit states in registers, the four state reads of a nibble issued together (their addresses depend on the
//                      nibble's bits, not on the states read); records arrive through an LDS transposition (coalesced loads).  The replay also CHECKS the sort it relies on (cells ascending inside a bin, events ascending
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
    uint32_t *pre2;       // [n_leaves][nblocks][1024]: exclusive prefix of the events per FINE bin (cell >> (log_cells - 10), or the cell itself
                          // for tables under 2^10 Cells): where a lane's range of Cells starts in the sorted array (k_slot_replay)
    const uint2 *st;      // [kStSize] {prob | next0 << 16, next1 | conf << 16}
    uint32_t *fault;      // the call's flag word 2 (order violations seen by the replay), or null
    uint8_t *dummy;       // >= 1 KiB sink for predicated-off stores
    int n_leaves;
    uint32_t *job_counter;               // k_slot_replay: next job (zeroed before the launch)
    uint32_t dbg;                        // timing experiments only (results wrong): 1 = no probability store
    uint32_t jobs_per_block;             // sum of leaf_w
    uint32_t leaf_w[W3_MAX_SLOT_LEAVES]; // wavefronts per (block, leaf): min(2^log_cells, 1024) / 64, at least 1, at most 4
    SlotLeaf leaf[W3_MAX_SLOT_LEAVES];   // (tbl_off unused)
};

#define W3_S2_CELL_SH 48u
#define W3_S2_TAG_SH 36u
#define W3_S2_NIB_SH 32u

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_slot_events(Slot2Args a) {
    __shared__ uint32_t s_hist[512];
    __shared__ uint32_t s_h2[1024];
    const uint32_t lane = threadIdx.x & 63u;
    const SlotLeaf &lf = a.leaf[blockIdx.y];
    const uint32_t order = lf.order, lshift = 64u - lf.log_cells;
    const uint32_t sh2 = lf.log_cells > 10u ? lf.log_cells - 10u : 0u;
    uint64_t *keys = a.keys_a + (uint64_t)blockIdx.y * 2u * a.n;
    for (uint32_t b = blockIdx.x; b < a.nblocks; b += gridDim.x) {
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
        const uint8_t *blk = a.in + off;
        const uint32_t first = window_head(off, 7u);
#pragma unroll
        for (int k = 0; k < 8; k++) s_hist[k * 64 + lane] = 0u;
#pragma unroll
        for (int k = 0; k < 16; k++) s_h2[k * 64 + lane] = 0u;
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
                __hip_atomic_fetch_add(&s_h2[cell >> sh2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        uint32_t *gh = a.hist + ((uint64_t)blockIdx.y * a.nblocks + b) * 512u;
#pragma unroll
        for (int k = 0; k < 8; k++) gh[k * 64 + lane] = s_hist[k * 64 + lane];
        {   // exclusive prefix of the fine bins: lane l owns bins 16 l .. 16 l + 15
            uint32_t cnt[16], sum = 0u;
#pragma unroll
            for (int k = 0; k < 16; k++) { cnt[k] = s_h2[16 * lane + k]; sum += cnt[k]; }
            uint32_t tot;
            uint32_t run = wave_excl_scan_u32(sum, &tot);
            uint32_t *gp = a.pre2 + ((uint64_t)blockIdx.y * a.nblocks + b) * 1024u + 16u * lane;
#pragma unroll
            for (int k = 0; k < 16; k++) { gp[k] = run; run += cnt[k]; }
        }
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
#define W3_S2_WAVES 8   // wavefronts per workgroup of k_slot_replay (they share the state table in LDS: 145 KiB per workgroup, one per CU)
#define W3_S2_CHUNK 8u  // records per lane and staging round

// One nibble on the Cell staged in LDS, tags and first-bit states of the four slots in REGISTERS (the Cell is private to the lane):
// slot lookup (or eviction), the four state steps with their reads issued together.  A slot's sector in LDS is cleared when the
// slot is first touched in this Cell (`valid`), not when the Cell is opened.
struct OpenCell { uint32_t t0, t1, t2, t3, f0, f1, f2, f3, valid; };
__device__ __forceinline__ void slot_nibble3(lds_u16 *cb, const lds_u64 *st, OpenCell &c, uint32_t tag, uint32_t nib, uint32_t p[4]) {
    // Cell::get_slot (hashmap.rs:42-63): compare the tags of slot 3, 2, 1, 0 in this order (an empty Cell's tags are 0)
    int id = tag == c.t3 ? 3 : tag == c.t2 ? 2 : tag == c.t1 ? 1 : tag == c.t0 ? 0 : -1;
    if (id < 0) {
        // miss (hashmap.rs:64-68 TODO; policy of w3_cm.h slot_select): victim = fewest observations in the slot's first-bit state,
        // candidates in the order 1, 0, 2, 3; tag stored, 15 states cleared
        uint32_t best = (uint32_t)(st[c.f1] >> 48); id = 1;
        uint32_t q = (uint32_t)(st[c.f0] >> 48); if (q < best) { best = q; id = 0; }
        q = (uint32_t)(st[c.f2] >> 48); if (q < best) { best = q; id = 2; }
        q = (uint32_t)(st[c.f3] >> 48); if (q < best) { best = q; id = 3; }
        c.t0 = id == 0 ? tag : c.t0; c.t1 = id == 1 ? tag : c.t1; c.t2 = id == 2 ? tag : c.t2; c.t3 = id == 3 ? tag : c.t3;
        c.valid &= ~(1u << id);
    }
    if (!((c.valid >> id) & 1u)) {   // first touch of this slot in this Cell (or just evicted): its 15 states start at 0
        typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
        lds_u32x4 *sec = (lds_u32x4 *)(cb + cx(16u * (uint32_t)id));
        u32x4 z; z.x = 0u; z.y = 0u; z.z = 0u; z.w = 0u;
        W3_LDS_FENCE();
        sec[0] = z;
        sec[64] = z;            // next chunk row: 64 lanes x 16 bytes further
        W3_LDS_FENCE();
        c.valid |= 1u << id;
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
    uint32_t nf = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t bit = (nib >> (3 - k)) & 1u;
        p[k] = (uint32_t)e[k] & 0xFFFFu;
        const uint32_t ns = bit ? ((uint32_t)(e[k] >> 32) & 0xFFFFu) : (((uint32_t)e[k] >> 16) & 0xFFFFu);
        cb[x[k]] = (uint16_t)ns;
        if (k == 0) nf = ns;   // the slot's first-bit state (node 0)
    }
    W3_LDS_FENCE();
    c.f0 = id == 0 ? nf : c.f0; c.f1 = id == 1 ? nf : c.f1; c.f2 = id == 2 ? nf : c.f2; c.f3 = id == 3 ? nf : c.f3;
}

// JOB = (block, leaf, w): wavefront w of the W_leaf (4; fewer for tables under 2^8 Cells) that share a (block, leaf); lane L = 64 w + lane replays
// the events of its fine bin(s) of Cells — contiguous in the sorted array, the start read from k_slot_events' prefix table — serially, with the open Cell
// in LDS.  Jobs are handed out in BLOCK-MAJOR order from one counter to a persistent grid (as k_rank_sorted does): with several wavefronts
// per (block, leaf) fewer streams are being scattered into at a time, so the 8-byte probability stores — four of them make a 32-byte
// sector, each from another Cell, i.e. another lane at another time — meet in the 256 MiB Infinity Cache instead of going to HBM as
// partial writes (one wavefront per (block, leaf), 2,048 streams live: the stores were 16 of the replay's 30 ms at enwik8 size; with 16
// wavefronts they cost 1 ms, but a lane then owns 16 Cells and the wavefront waits for its busiest lane: see twophase_predict_b for the numbers).
// Shorter jobs also mean that a lane stuck with a hot Cell (few distinct contexts: one Cell takes most of a block's events) holds up one
// wavefront's lanes for their share, not for a whole block's.  Records reach the lanes through an LDS transposition: a lane's records are consecutive in
// memory, so 8 lanes fetch the next W3_S2_CHUNK records of ONE lane's range in one coalesced access and each lane then reads its own
// records back from LDS (one load instruction per eight lanes' chunks instead of one fully divergent load per event).
__global__ void __launch_bounds__(64 * W3_S2_WAVES) k_slot_replay(Slot2Args a, int two_passes) {
    __shared__ uint2 s_st[kStSize];
    __shared__ u32x4 s_cell[W3_S2_WAVES][8][64];   // [wave][chunk][lane]: the open Cell of every lane
    __shared__ uint32_t s_lo[W3_S2_WAVES][64], s_n[W3_S2_WAVES][64];
    __shared__ uint64_t s_rec[W3_S2_WAVES][W3_S2_CHUNK * 65u];   // [round r][lane] (+1 per row: the staging writes would hit one bank)
    for (uint32_t i = threadIdx.x; i < (uint32_t)kStSize; i += 64u * W3_S2_WAVES) s_st[i] = a.st[i];
    __syncthreads();
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) u32x2 g_uint2;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    lds_u16 *cb = (lds_u16 *)&s_cell[wv][0][lane];
    const lds_u64 *st = (const lds_u64 *)&s_st[0];
    uint64_t *rec = s_rec[wv];
    g_uint2 *sink = (g_uint2 *)(a.dummy + 8u * lane);
    const uint32_t tl = lane >> 3, rl = lane & 7u;   // staging geometry: load instruction g fetches record rl of the chunks of owners 8 g + tl
    uint32_t bad = 0u;
    const uint32_t njobs = a.nblocks * a.jobs_per_block;
    for (;;) {   // (no barrier below: the wavefronts of a workgroup take jobs independently)
        uint32_t job = 0;
        if (lane == 0) job = atomicAdd(a.job_counter, 1u);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= njobs) break;
        const uint32_t b = job / a.jobs_per_block;
        uint32_t rj = job % a.jobs_per_block, lfi = 0u;
        while (lfi + 1u < (uint32_t)a.n_leaves && rj >= a.leaf_w[lfi]) { rj -= a.leaf_w[lfi]; lfi++; }
        const SlotLeaf &lf = a.leaf[lfi];
        const uint32_t W = a.leaf_w[lfi], lc = lf.log_cells;
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint32_t len = (uint32_t)((a.n - off) < a.block_size ? (a.n - off) : a.block_size);
        const uint32_t nev = 2u * len;
        const uint64_t *keys = (two_passes ? a.keys_a : a.keys_b) + (uint64_t)lfi * 2u * a.n + 2u * off;
        g_uint2 *Pout = (g_uint2 *)(lf.P + off);
        // this lane's range of fine bins (k_slot_events' prefix table), i.e. of Cells and of sorted records
        const uint32_t L = rj * 64u + lane, lanes_total = 64u * W;
        const uint32_t sh2 = lc > 10u ? lc - 10u : 0u, nbins = 1u << (lc - sh2);
        const uint32_t b_lo = (uint32_t)(((uint64_t)L * nbins) / lanes_total), b_hi = (uint32_t)(((uint64_t)(L + 1u) * nbins) / lanes_total);
        const uint32_t c_lo = b_lo << sh2, c_hi = b_hi << sh2;
        const uint32_t *pre = a.pre2 + ((uint64_t)lfi * a.nblocks + b) * 1024u;
        const uint32_t p_lo = pre[min(b_lo, nbins - 1u)], p_hi = b_hi < nbins ? pre[b_hi] : nev;
        const uint32_t bs = p_lo, bc = b_hi > b_lo ? p_hi - p_lo : 0u;
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        s_lo[wv][lane] = bs; s_n[wv][lane] = bc;
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        uint32_t maxc = bc;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) maxc = max(maxc, (uint32_t)__shfl_xor((int)maxc, d, 64));
        maxc = __builtin_amdgcn_readfirstlane(maxc);
        if (maxc == 0u) continue;
        uint32_t os[8], oc[8];   // the owners' ranges for the staging loads
#pragma unroll
        for (uint32_t g = 0; g < 8u; g++) { os[g] = s_lo[wv][8u * g + tl]; oc[g] = s_n[wv][8u * g + tl]; }
        OpenCell c; c.t0 = c.t1 = c.t2 = c.t3 = 0u; c.f0 = c.f1 = c.f2 = c.f3 = 0u; c.valid = 0u;
        uint32_t cur_cell = 0xFFFFFFFFu, prev_e = 0u;
        uint64_t kn[8];
#pragma unroll
        for (uint32_t g = 0; g < 8u; g++) kn[g] = keys[oc[g] ? os[g] + min(rl, oc[g] - 1u) : 0u];
        for (uint32_t k0 = 0; k0 < maxc; k0 += W3_S2_CHUNK) {
            // stage the chunk: record r of lane t at rec[r * 65 + t]
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t g = 0; g < 8u; g++) rec[rl * 65u + 8u * g + tl] = kn[g];
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t g = 0; g < 8u; g++) kn[g] = keys[oc[g] ? os[g] + min(k0 + W3_S2_CHUNK + rl, oc[g] - 1u) : 0u];   // the next chunk
#pragma unroll
            for (uint32_t j = 0; j < W3_S2_CHUNK; j++) {
                const bool act = k0 + j < bc;
                const uint64_t key = rec[j * 65u + lane];
                const uint32_t cell = (uint32_t)(key >> W3_S2_CELL_SH), e = (uint32_t)key;
                const uint32_t tag = (uint32_t)(key >> W3_S2_TAG_SH) & 0xFFFu, nib = (uint32_t)(key >> W3_S2_NIB_SH) & 15u;
                uint32_t pq[4] = {0u, 0u, 0u, 0u};
                if (act) {
                    if (cell != cur_cell) {   // the previous Cell's events are through: this one starts empty (a fresh HashMap is all zeros)
                        // the sort this replay relies on: the lane's cells ascend inside its range ...
                        bad |= ((cur_cell != 0xFFFFFFFFu && cell < cur_cell) || cell < c_lo || cell >= c_hi) ? 1u : 0u;
                        c.t0 = c.t1 = c.t2 = c.t3 = 0u; c.f0 = c.f1 = c.f2 = c.f3 = 0u; c.valid = 0u;
                        cur_cell = cell;
                    } else bad |= e <= prev_e ? 1u : 0u;   // ... and inside a cell the events keep their time order (the partition is stable)
                    prev_e = e;
                    slot_nibble3(cb, st, c, tag, nib, pq);
                }
                u32x2 v; v.x = pq[0] | (pq[1] << 16); v.y = pq[2] | (pq[3] << 16);
                g_uint2 *dst = (act && !(a.dbg & 1u)) ? Pout + e : sink;
                *dst = v;
            }
        }
    }
    if (bad && a.fault) atomicAdd(a.fault, 1u);
}

}  // namespace w3
