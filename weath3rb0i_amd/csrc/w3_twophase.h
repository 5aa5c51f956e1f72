// w3_twophase.h — host orchestration of the two-phase encoder:
//   predict (w3_predict.h): all leaves' Counter probabilities for every step, merged into P
//   code    (w3_coder.h)  : lane-per-block arithmetic coder over P
// Covered specs: every leaf is FrozenModel, or has alignment_bits == 3 and
//   H = bits-3 <= 8 (any history), or H in {16, 24} with raw history.
// Anything else runs on the generic path (w3_generic.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "w3_apm.h"
#include "w3_coder.h"
#include "w3_coder4.h"
#include "w3_coder5.h"
#include "w3_predict.h"
#include "w3_predict_wave.h"
#include "w3_slot.h"
#include "w3_slot2.h"
#include "w3_spec.h"

// Tuning / timing-experiment hooks read from the environment exist only in -DW3_TUNING builds: the shipped library never
// calls getenv.  (Some of them make the kernels skip work, i.e. produce wrong results.)
static inline const char *w3_tune_env(const char *name) {
#ifdef W3_TUNING
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// W3_OPT_VARIANT bits: alternative (bit-exact) implementations that the tests cross-check against the default ones
enum { W3_VAR_NO_LDS_ATOMICS = 1, W3_VAR_PARTITION4 = 2, W3_VAR_NO_CHAINED_PARTITION = 4, W3_VAR_CM_UNSTAGED = 8, W3_VAR_NO_SIDE_STREAM = 16,
       W3_VAR_INJECT_LDS_FAULT = 32 /* tests: corrupt one LDS-add round per block, the sampled verification must catch it */,
       W3_VAR_HALF_CU = 64    /* synchronous calls too run the half-CU kernel shapes of the submit / wait pipeline (w3_predict.h) */,
       W3_VAR_FULL_CU = 128   /* w3_encode_submit keeps the plain kernel shapes (experiments: what the shapes are worth) */,
       W3_VAR_SLOT_TABLE = 256  /* slot-state leaves always on k_slot (hash map in HBM, lane per block) */,
       W3_VAR_SLOT_SORTED = 512 /* slot-state leaves always on the sorted replay of w3_slot2.h (default: by block count) */,
       W3_VAR_DECODE_LANE = 1024 /* decode with the lane-per-block kernels only (k_generic_nl / k_cm_nl), not k_decode_spec */ };

#define W3_SLOT_SORTED_MAX_BLOCKS 7000u   // below: slot-state leaves by sorted replay (w3_slot2.h), from here on k_slot

// event slots (pairs: ev[2 * slot], ev[2 * slot + 1]) of one encode
enum { W3_EV_PREDICT = 0, W3_EV_CODER = 1, W3_EV_PACK = 2, W3_EV_TOTAL = 3, W3_EV_APM = 4, W3_EV_SLOT = 5, W3_EV_ACHASH = 6,
       W3_EV_PART0 = 7 /* .. 10: partition pass of wide leaf w */, W3_EV_RANK0 = 11 /* .. 14: rank kernel of wide leaf w */, W3_EV_SMALL = 15,
       W3_EV_SLOTS = 16 };
#define W3_NEV (2 * W3_EV_SLOTS)

struct TwoPhaseWs {
    void *P = nullptr, *keys = nullptr, *perm = nullptr, *redo = nullptr, *streams = nullptr, *rec = nullptr, *splits = nullptr;
    size_t P_cap = 0, keys_cap = 0, perm_cap = 0, redo_cap = 0, streams_cap = 0, rec_cap = 0, splits_cap = 0;
    w3::MixArgs mix{};         // leaf streams of the last predict (sources of k_mix / k_coder_x3)
    bool P_valid = false;      // ws.P holds the merged stream of the last predict
    void *dbg = nullptr;       // 8 x u64 phase stamps of the last wide predict kernel (W3_OPT_DEBUG_STAMPS)
    int debug_stamps = 0;
    bool achash_timed = false; // the last predict call recorded ev[12]/ev[13]
    hipEvent_t ev_pred_done = nullptr;   // two-stream form of twophase_encode: the predict phase is through (the code stream waits for it)
    int lds_order = -1;        // k_lds_order_selftest: -1 not run yet, 1 = returning LDS adds are lane-ordered (atomic rounds allowed), 0 = not
    const int16_t *stretch = nullptr;   // APM LUTs (device; owned by the ctx)
    const uint16_t *squash = nullptr;
    const uint2 *st = nullptr;          // NaiveStateTable rows for the slot-state leaves (device; owned by the ctx)
    const w3_huff_table *huff = nullptr;   // HuffHistory table sets of the current call's spec (device; owned by the ctx)
    uint32_t *out_bits = nullptr;          // [nb of this range] ACStats bit counts (device; owned by the ctx), or null
    // wide Counter leaves: their k_partition passes run on a side stream beside the time-ordered leaves' kernels
    void *rec_w[4] = {nullptr, nullptr, nullptr, nullptr}, *perm_w[4] = {nullptr, nullptr, nullptr, nullptr}, *splits_w[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t rec_w_cap[4] = {0, 0, 0, 0}, perm_w_cap[4] = {0, 0, 0, 0}, splits_w_cap[4] = {0, 0, 0, 0};
    int wide1_slot = -1;                // rec_w / splits_w index of an Order1-shaped leaf of the last predict (records sorted by c1)
    hipStream_t side = nullptr;
    // what the first half of the predict phase (twophase_predict_a: keys, partition passes, time-ordered leaves) leaves for the second
    // (twophase_predict_b: rank kernels of the wide leaves, slot-state leaves, merge)
    struct PredictState {
        bool forked = false, lds_atomics = false;
        int n_def = 0, n_small_def = 0, n_live = 0;
        struct Deferred { w3::PredictArgs pa; int cls; uint32_t grid_rank; } deferred[4];
        uint64_t bytes = 0, slot_stride = 0;
        w3::SlotArgs sa;
    } pst;
    uint32_t *order_fault = nullptr;   // the call's flag word 2 (k_slot_replay reports a sort that lost its order there), or null
    uint32_t *apm_oob = nullptr;   // -DW3_TUNING builds: the call's flag word 3 (W3_APM_CHECK_STORE), else null
    uint32_t tune = 0;         // W3_OPT_TUNE: scheduling experiments (bit 1: k_apm0 padded to one workgroup per CU in the half-CU shapes)
    bool half_cu = false;      // half-CU kernel shapes (w3_predict.h W3_HALF_CU_LDS, k_coder_x5): the call shares every CU with another call's stage
    int n_wide = 0; bool small_timed = false;   // what the last predict recorded events for (W3_EV_PART0.., W3_EV_RANK0.., W3_EV_SMALL)
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr}, ev_small = nullptr;
    void *achash_lut = nullptr;         // k_achash_lut: [65536][8] coder states (8 MiB)
    size_t achash_lut_cap = 0;
    void *dummy = nullptr;              // 256-byte sink for predicated-off stores (w3_apm.h)
    size_t dummy_cap = 0;
    void *slot_tables = nullptr;        // per-lane HashMaps of the slot-state leaves
    size_t slot_tables_cap = 0;
    void *keys32 = nullptr; size_t keys32_cap = 0;            // [8 n] u32 hashes of a wide OrderNEntropy leaf (k_achash32 / k_huffkeys<true>)
    void *slot_keys = nullptr; size_t slot_keys_cap = 0;      // w3_slot2.h: [2][n_leaves][2 n] u64 event records (ping-pong) + [n_leaves][nb][512] digit counts
    bool slot_sorted = false;                                 // what the last predict ran its slot leaves on
    void *wave_tables = nullptr; size_t wave_tables_cap = 0;  // k_predict_wave: one Counter table per resident wavefront
    void *huff_redo = nullptr;          // k_huffkeys: per-block "recompute serially" flags
    size_t huff_redo_cap = 0;
    int coder_mode = 0;        // 0 = k_coder_x4 (mix + asm recurrence + output waves), 1 = k_coder_fast, 2 = k_coder only, 3 = k_coder_x2, 4 = k_coder_x3, 5 = k_coder_x5 (x4 with 72 KiB rings)
    uint32_t acc_limit = 46;   // test hook: lower values force the fast coder's fallback
    uint32_t variant = 0;      // W3_VAR_* (W3_OPT_VARIANT)
    uint32_t slot_budget_mb = 0;   // W3_OPT_SLOT_BUDGET_MB: cap on the slot leaves' hash-map batch (0 = from the free device memory)
    uint32_t verify_calls = 0; // rotates the verification's sample
    uint32_t fault_block = 0xFFFFFFFFu;   // W3_OPT_FAULT_BLOCK (with W3_VAR_INJECT_LDS_FAULT)
    int verify = 1;            // W3_OPT_VERIFY: sampled re-prediction with ballot rounds after every predict phase that used LDS-add rounds
    TwoPhaseWs *vws = nullptr; // workspace of that re-prediction (owned)
    void *vin = nullptr; size_t vin_cap = 0;   // the sampled blocks, gathered
    bool used_lds_atomics = false;              // the last predict ran LDS-add rounds in some kernel
    hipStream_t vstream = nullptr; hipEvent_t ev_v0 = nullptr, ev_v1 = nullptr;   // the re-prediction runs beside the APM and coder kernels
    bool ext_streams = false;   // side / vstream belong to the context (w3hip.hip hands them from context to context): not destroyed here
    void release() {
        if (vws) { vws->release(); delete vws; vws = nullptr; }
        if (vin) (void)hipFree(vin);
        vin = nullptr; vin_cap = 0;
        if (vstream && !ext_streams) (void)hipStreamDestroy(vstream);
        if (ev_v0) (void)hipEventDestroy(ev_v0);
        if (ev_v1) (void)hipEventDestroy(ev_v1);
        vstream = nullptr; ev_v0 = ev_v1 = nullptr;
        if (P) (void)hipFree(P);
        if (keys) (void)hipFree(keys);
        if (perm) (void)hipFree(perm);
        if (redo) (void)hipFree(redo);
        for (int w = 0; w < 4; w++) {
            if (rec_w[w]) (void)hipFree(rec_w[w]);
            if (perm_w[w]) (void)hipFree(perm_w[w]);
            if (splits_w[w]) (void)hipFree(splits_w[w]);
            rec_w[w] = perm_w[w] = splits_w[w] = nullptr; rec_w_cap[w] = perm_w_cap[w] = splits_w_cap[w] = 0;
        }
        if (side && !ext_streams) (void)hipStreamDestroy(side);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (int w = 0; w < 4; w++) { if (ev_join[w]) (void)hipEventDestroy(ev_join[w]); ev_join[w] = nullptr; }
        if (ev_small) (void)hipEventDestroy(ev_small);
        ev_small = nullptr;
        side = nullptr; ev_fork = nullptr;
        if (ev_pred_done) (void)hipEventDestroy(ev_pred_done);
        ev_pred_done = nullptr;
        if (achash_lut) (void)hipFree(achash_lut);
        achash_lut = nullptr; achash_lut_cap = 0;
        if (dummy) (void)hipFree(dummy);
        dummy = nullptr; dummy_cap = 0;
        if (slot_tables) (void)hipFree(slot_tables);
        slot_tables = nullptr; slot_tables_cap = 0;
        if (keys32) (void)hipFree(keys32);
        keys32 = nullptr; keys32_cap = 0;
        if (slot_keys) (void)hipFree(slot_keys);
        slot_keys = nullptr; slot_keys_cap = 0;
        if (wave_tables) (void)hipFree(wave_tables);
        wave_tables = nullptr; wave_tables_cap = 0;
        if (huff_redo) (void)hipFree(huff_redo);
        huff_redo = nullptr; huff_redo_cap = 0;
        if (streams) (void)hipFree(streams);
        if (rec) (void)hipFree(rec);
        if (splits) (void)hipFree(splits);
        rec = splits = nullptr; rec_cap = splits_cap = 0;
        if (dbg) (void)hipFree(dbg);
        dbg = nullptr;
        P = keys = perm = redo = streams = nullptr;
        P_cap = keys_cap = perm_cap = redo_cap = streams_cap = 0;
    }
};

static inline uint64_t tp_next_pow2(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }

static inline int tp_ensure(void *&p, size_t &cap, size_t bytes, std::string &err) {
    if (p && bytes <= cap) return W3_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 256));
    if (e != hipSuccess) { p = nullptr; (void)hipGetLastError(); err = "hipMalloc(" + std::to_string(bytes) + ") for the two-phase workspace failed"; return W3_E_NOMEM; }
    cap = std::max<size_t>(bytes, 256);
    return W3_OK;
}

enum { LEAF_FROZEN = 0, LEAF_SMALL = 1, LEAF_SMALL_AC = 2, LEAF_WIDE1 = 3, LEAF_WIDE2 = 4, LEAF_SLOT = 5, LEAF_SMALL_HUFF = 6,
       LEAF_WAVE = 7 /* any other Counter-table leaf: wave per block, table in HBM (w3_predict_wave.h) */, LEAF_NONE = -1 };

static inline int leaf_class(const w3_node &nd) {
    if (nd.kind == W3_NODE_SLOT_STATE) return LEAF_SLOT;
    if (nd.frozen) return LEAF_FROZEN;
    if (nd.align != 3 || nd.bits < 3) return LEAF_WAVE;
    const int H = nd.bits - 3;
    if (nd.history == W3_HIST_HUFF) return H <= 8 ? LEAF_SMALL_HUFF : LEAF_WAVE;
    if (H <= 8) return nd.history == W3_HIST_AC ? LEAF_SMALL_AC : LEAF_SMALL;
    if (nd.history == W3_HIST_AC) return LEAF_WAVE;
    if (H == 16) return LEAF_WIDE1;
    if (H == 24) return LEAF_WIDE2;
    return LEAF_WAVE;
}

static inline bool twophase_supported(const ParsedSpec &ps, size_t block_size, size_t n) {
    if (block_size > (1u << 24) || n < 8) return false;   // the window loads read 4 (k_predict_wave: 8) bytes at once
    int n_slot = 0;
    for (int l = 0; l < ps.n_leaves; l++) n_slot += ps.leaf[l].kind == W3_NODE_SLOT_STATE;
    if (n_slot > W3_MAX_SLOT_LEAVES) return false;
    for (int l = 0; l < ps.n_leaves; l++)
        if (leaf_class(ps.leaf[l]) == LEAF_NONE) return false;
    return true;
}

// waves per half-CU workgroup (NW * per-wave LDS <= W3_HALF_CU_LDS): k_predict_small 10,240 B, k_partition8 20,480 B, k_rank_sorted 12,288 B
#define W3_NW_SMALL 8
#define W3_NW_PART 4
#define W3_NW_RANK 6
#define W3_NW_RANK8 8      // with 4-round operand batches (10 KiB per wavefront): eight wavefronts in the half of a CU (W3_OPT_TUNE bit 15)
#define W3_HALF_CU_GRID 256u   // one workgroup per CU

template <bool KEYS, int NW = 1>
static inline void launch_small(int H, dim3 grid, hipStream_t s, const w3::PredictArgs &pa) {
    const dim3 blk(64 * NW);
    switch (H) {
    case 0: hipLaunchKernelGGL((w3::k_predict_small<0, KEYS, NW>), grid, blk, 0, s, pa); break;
    case 1: hipLaunchKernelGGL((w3::k_predict_small<1, KEYS, NW>), grid, blk, 0, s, pa); break;
    case 2: hipLaunchKernelGGL((w3::k_predict_small<2, KEYS, NW>), grid, blk, 0, s, pa); break;
    case 3: hipLaunchKernelGGL((w3::k_predict_small<3, KEYS, NW>), grid, blk, 0, s, pa); break;
    case 4: hipLaunchKernelGGL((w3::k_predict_small<4, KEYS, NW>), grid, blk, 0, s, pa); break;
    case 5: hipLaunchKernelGGL((w3::k_predict_small<5, KEYS, NW>), grid, blk, 0, s, pa); break;
    case 6: hipLaunchKernelGGL((w3::k_predict_small<6, KEYS, NW>), grid, blk, 0, s, pa); break;
    case 7: hipLaunchKernelGGL((w3::k_predict_small<7, KEYS, NW>), grid, blk, 0, s, pa); break;
    default: hipLaunchKernelGGL((w3::k_predict_small<8, KEYS, NW>), grid, blk, 0, s, pa); break;
    }
}
// time-ordered Counter leaf in the shape the workspace asks for
static inline void launch_small_shaped(const TwoPhaseWs &ws, int H, uint32_t nb, hipStream_t s, const w3::PredictArgs &pa) {
    if (ws.half_cu && (ws.tune & 32u)) launch_small<false, W3_NW_SMALL>(H, dim3(std::min<uint32_t>((nb + W3_NW_SMALL - 1) / W3_NW_SMALL, 2 * W3_HALF_CU_GRID)), s, pa);
    else launch_small<false>(H, dim3(std::min<uint32_t>(nb, 256 * 20)), s, pa);
}

// Runs the predict kernels of every leaf; *d_P receives the merged stream.
static inline int twophase_mix(TwoPhaseWs &ws, hipStream_t s, size_t n, std::string &err) {
    if (ws.P_valid) return W3_OK;
    int rc = tp_ensure(ws.P, ws.P_cap, n * 16, err);
    if (rc) return rc;
    ws.mix.P = (uint4 *)ws.P; ws.mix.n = n;
    hipLaunchKernelGGL(w3::k_mix, dim3(256 * 8), dim3(256), 0, s, ws.mix);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("mix launch: ") + hipGetErrorString(e); return W3_E_HIP; }
    ws.P_valid = true;
    return W3_OK;
}

// Runs k_lds_order_selftest once per context and replays it on the host (see atomic_round in w3_predict.h).
static inline bool twophase_lds_order_ok(TwoPhaseWs &ws, hipStream_t s) {
    if (ws.lds_order >= 0) return ws.lds_order == 1;
    ws.lds_order = 0;
    if (ws.variant & W3_VAR_NO_LDS_ATOMICS) return false;
    const uint32_t waves = 64, rounds = 32, n = waves * rounds * 64;
    uint32_t *d_old = nullptr;
    if (hipMalloc(&d_old, (size_t)n * 4) != hipSuccess) { (void)hipGetLastError(); return false; }
    std::vector<uint32_t> old(n), t(2048);
    hipLaunchKernelGGL(w3::k_lds_order_selftest, dim3(waves), dim3(64), 0, s, d_old);
    bool ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(old.data(), d_old, (size_t)n * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
              hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(d_old);
    if (!ok) { (void)hipGetLastError(); return false; }
    for (uint32_t w = 0; w < waves && ok; w++) {
        std::fill(t.begin(), t.end(), 0u);
        for (uint32_t r = 0; r < rounds && ok; r++)
            for (uint32_t l = 0; l < 64; l++) {
                const uint32_t h = w3::lds_order_hash(w, r, l), k = w3::lds_order_key(w, h);
                if (old[(w * rounds + r) * 64 + l] != t[k]) { ok = false; break; }
                t[k] += (h >> 20) & 1u ? 0x10000u : 1u;
            }
    }
    ws.lds_order = ok ? 1 : 0;
    return ok;
}

// need_P: also merge the leaves' streams into ws.P (k_mix).  The default coder (k_coder_x3) mixes on the fly
// and needs no P for up to 4 live leaves.
// The predict phase in two halves (the submit / wait pipeline puts the previous call's APM stage between them):
//   a: context keys, partition passes of the wide leaves, time-ordered leaves — kernels that fill every CU's LDS
//   b: rank kernels of the wide leaves (bound by their scattered stores: the coder of the previous call runs beside them),
//      slot-state leaves, merge
// join_all: s also waits for the side stream's kernels (the caller wants the whole first half behind one event)
static inline int twophase_predict_a(TwoPhaseWs &ws, hipStream_t s, const ParsedSpec &ps, const uint8_t *d_in, size_t n, size_t block_size,
                                     uint32_t nb, hipEvent_t *ev, std::string &err, bool join_all = false) {
    int rc = W3_OK;
    ws.P_valid = false;
    ws.achash_timed = false;
    const bool lds_atomics = twophase_lds_order_ok(ws, s);
    ws.used_lds_atomics = false;
    const uint32_t grid_small = std::min<uint32_t>(nb, 256 * 20);   // (KEYS leaves; the plain ones: launch_small_shaped)
    const uint32_t grid_wide = std::min<uint32_t>(nb, 256 * 16);
    ws.n_wide = 0; ws.small_timed = false;
    bool need_keys = false, need_perm = false;
    for (int l = 0; l < ps.n_leaves; l++) {
        int c = leaf_class(ps.leaf[l]);
        need_keys |= c == LEAF_SMALL_AC || c == LEAF_SMALL_HUFF;
        need_perm |= c == LEAF_WIDE1 || c == LEAF_WIDE2;
    }
    if (need_keys && (rc = tp_ensure(ws.keys, ws.keys_cap, n * 8, err))) return rc;
    if (need_perm) {
        int nw = 0;
        for (int l = 0; l < ps.n_leaves; l++) { const int c = leaf_class(ps.leaf[l]); nw += c == LEAF_WIDE1 || c == LEAF_WIDE2; }
        if (nw > 4) { err = "more than 4 wide Counter leaves"; return W3_E_UNSUPPORTED; }
        for (int w = 0; w < nw; w++) {
            if ((rc = tp_ensure(ws.perm_w[w], ws.perm_w_cap[w], (size_t)grid_wide * 2 * block_size * 8, err))) return rc;
            if ((rc = tp_ensure(ws.rec_w[w], ws.rec_w_cap[w], n * 8, err))) return rc;
            if ((rc = tp_ensure(ws.splits_w[w], ws.splits_w_cap[w], (size_t)nb * (W3_SLICES + 1) * 4 + 64, err))) return rc;
        }
        if (!ws.side || !ws.ev_fork) {
            bool ok = (ws.side || hipStreamCreateWithFlags(&ws.side, hipStreamNonBlocking) == hipSuccess) && hipEventCreateWithFlags(&ws.ev_fork, hipEventDisableTiming) == hipSuccess;
            for (int w = 0; w < 4 && ok; w++) ok = hipEventCreateWithFlags(&ws.ev_join[w], hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&ws.ev_small, hipEventDisableTiming) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); err = "side stream creation failed"; return W3_E_HIP; }
        }
    }

    // FrozenModel leaves never adapt: p == 32768, distance 0.  They can never beat a trained leaf and tie
    // only when every leaf says 32768, so they matter only if ALL leaves are frozen.
    int live[W3_MAX_LEAVES], n_live = 0;
    for (int l = 0; l < ps.n_leaves; l++)
        if (leaf_class(ps.leaf[l]) != LEAF_FROZEN) live[n_live++] = l;
    if (n_live > 1 && (rc = tp_ensure(ws.streams, ws.streams_cap, (size_t)n_live * n * 16, err))) return rc;
    if (n_live <= 1 && (rc = tp_ensure(ws.P, ws.P_cap, n * 16, err))) return rc;

    if ((rc = tp_ensure(ws.dummy, ws.dummy_cap, 2048, err))) return rc;
    if (ev) (void)hipEventRecord(ev[2 * W3_EV_PREDICT], s);
    // fork: the partition passes of the wide leaves go to the side stream (they are bound by scattered line requests, the
    // time-ordered kernels by VALU issue); the rank kernels follow on the main stream after the join
    const bool forked = need_perm && !(ws.variant & W3_VAR_NO_SIDE_STREAM) && !(ws.half_cu && (ws.tune & 32u));
    hipStream_t sp = forked ? ws.side : s;
    if (forked) { (void)hipEventRecord(ws.ev_fork, s); (void)hipStreamWaitEvent(ws.side, ws.ev_fork, 0); }
    ws.wide1_slot = -1;
    TwoPhaseWs::PredictState &st = ws.pst;
    auto &deferred = st.deferred;
    int n_def = 0;
    w3::PredictArgs small_def[8];   // time-ordered table leaves held back until the partition passes are queued
    int n_small_def = 0;
    uint64_t bytes = 0;
    w3::MixArgs &ma = ws.mix;
    memset(&ma, 0, sizeof ma);
    w3::SlotArgs &sa = st.sa;
    memset(&sa, 0, sizeof sa);
    uint64_t slot_stride = 0;
    bool achash_timed = false;
    if (n_live == 0) {
        hipLaunchKernelGGL(w3::k_fill_half, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint4 *)ws.P, (uint64_t)n);
        bytes += n * 16;
        ma.src[0] = (const uint4 *)ws.P; ma.n_src = 1;
        ws.P_valid = true;
    }
    for (int k = 0; k < n_live; k++) {
        const w3_node &nd = ps.leaf[live[k]];
        const int c = leaf_class(nd);
        if (lds_atomics && c != LEAF_SLOT && c != LEAF_WAVE) ws.used_lds_atomics = true;   // the sorted / time-ordered Counter kernels run LDS-add rounds
        w3::PredictArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.in = d_in; pa.n = n; pa.block_size = (uint32_t)block_size; pa.nblocks = nb;
        pa.P = n_live == 1 ? (uint4 *)ws.P : (uint4 *)ws.streams + (size_t)k * n;   // a single leaf writes P directly
        ma.src[k] = pa.P;
        if (c == LEAF_SLOT) {   // all slot-state leaves run in one k_slot launch after the Counter leaves
            w3::SlotLeaf &sl = sa.leaf[sa.n_leaves++];
            sl.order = nd.bits; sl.log_cells = nd.log_cells; sl.tbl_off = slot_stride; sl.P = pa.P;
            slot_stride += W3_CELL_STRIDE << nd.log_cells;
            continue;
        }
        if (c == LEAF_WAVE) {
            // wave per block, time order, Counter table in HBM (w3_predict_wave.h)
            const uint64_t hash_slots = std::max<uint64_t>(1024, tp_next_pow2(16ull * block_size));
            const uint64_t hash_bytes = 8ull * hash_slots + 16ull, direct_bytes = std::max<uint64_t>(4ull << nd.bits, 16ull);
            w3::WaveArgs wa;
            memset(&wa, 0, sizeof wa);
            wa.in = d_in; wa.n = n; wa.block_size = (uint32_t)block_size; wa.nblocks = nb; wa.P = (uint16_t *)pa.P;
            wa.align = nd.align; wa.hist_mask = (uint32_t)((1ull << (nd.bits - nd.align)) - 1ull);
            wa.use_hash = (nd.bits >= 32 || direct_bytes > hash_bytes) ? 1u : 0u;
            wa.hash_slots = (uint32_t)hash_slots;
            wa.table_stride = wa.use_hash ? hash_bytes : direct_bytes;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); err = "hipMemGetInfo failed"; return W3_E_HIP; }
            const uint64_t budget = std::min<uint64_t>(((uint64_t)free_b + ws.wave_tables_cap) / 2, 80ull << 30);
            const uint64_t waves = std::min<uint64_t>(std::min<uint64_t>(nb, 256 * 32), budget / wa.table_stride);   // latency-bound: as many wavefronts as the chip holds
            if (waves == 0) { err = "the Counter table of one block (" + std::to_string(wa.table_stride) + " B) exceeds the device budget"; return W3_E_NOMEM; }
            if ((rc = tp_ensure(ws.wave_tables, ws.wave_tables_cap, (size_t)(waves * wa.table_stride), err))) return rc;
            wa.tables = (uint8_t *)ws.wave_tables;
            const bool keyed = nd.history == W3_HIST_AC || nd.history == W3_HIST_HUFF;
            if (keyed) {
                if ((rc = tp_ensure(ws.keys32, ws.keys32_cap, n * 32, err))) return rc;
                wa.keys32 = (const uint32_t *)ws.keys32;
                if (nd.history == W3_HIST_AC) {
                    w3::HashArgs ha;
                    memset(&ha, 0, sizeof ha);
                    ha.max_bits = nd.max_bits; ha.hmask = 0xFFu;
                    memcpy(ha.table, nd.table, sizeof ha.table);
                    if ((rc = tp_ensure(ws.achash_lut, ws.achash_lut_cap, (size_t)(8u << W3_ACHASH_LUT_BITS) * 18, err))) return rc;
                    ha.lut = (uint4 *)ws.achash_lut;
                    ha.lut_key = (uint16_t *)((uint8_t *)ws.achash_lut + (size_t)(8u << W3_ACHASH_LUT_BITS) * 16);
                    hipLaunchKernelGGL(w3::k_achash_lut, dim3((8u << W3_ACHASH_LUT_BITS) / 256), dim3(256), 0, s, ha);
                    w3::Hash32Args h3;
                    memset(&h3, 0, sizeof h3);
                    h3.in = d_in; h3.n = n; h3.block_size = (uint32_t)block_size; h3.max_bits = nd.max_bits;
                    memcpy(h3.table, nd.table, sizeof h3.table);
                    h3.lut = (const uint4 *)ws.achash_lut; h3.keys32 = (uint32_t *)ws.keys32;
                    hipLaunchKernelGGL(w3::k_achash32, dim3((unsigned)std::min<uint64_t>(((uint64_t)n * 8 + 255) / 256, 1u << 22)), dim3(256), 0, s, h3);   // (grid-stride: < 2^32 work-items)
                } else {
                    if (!ws.huff) { err = "HuffHistory tables not staged"; return W3_E_HIP; }
                    w3::HuffKeyArgs hk;
                    memset(&hk, 0, sizeof hk);
                    hk.in = d_in; hk.n = n; hk.block_size = (uint32_t)block_size; hk.hmask = 0xFFFFFFFFu;
                    hk.tb = ws.huff + nd.reserved; hk.keys32 = (uint32_t *)ws.keys32;
                    if ((rc = tp_ensure(ws.huff_redo, ws.huff_redo_cap, (size_t)nb * 4, err))) return rc;
                    hk.redo = (uint32_t *)ws.huff_redo; hk.nblocks = nb;
                    (void)hipMemsetAsync(ws.huff_redo, 0, (size_t)nb * 4, s);
                    hipLaunchKernelGGL(w3::k_huffkeys<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, hk);
                    hipLaunchKernelGGL(w3::k_huffkeys_fix<true>, dim3((nb + 63u) / 64u), dim3(64), 0, s, hk);
                }
                hipLaunchKernelGGL(w3::k_predict_wave<true>, dim3((unsigned)waves), dim3(64), 0, s, wa);
                bytes += n * (1 + 32 + 32);
            } else {
                hipLaunchKernelGGL(w3::k_predict_wave<false>, dim3((unsigned)waves), dim3(64), 0, s, wa);
            }
            bytes += n * (8 + 16 + 8 * 16);   // per step: 8 input bytes read, 2 written, one 8-byte slot read and written (as 32-byte sectors: reported by the PMC passes)
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { err = std::string("wave predict launch: ") + hipGetErrorString(e); return W3_E_HIP; }
            continue;
        }
        pa.hbits = nd.bits - 3;
        pa.sink = (uint4 *)ws.dummy;
        pa.maxseg = W3_ATOMIC_MAXSEG;
        if (const char *ev_ = w3_tune_env("W3_ATOMIC_MAXSEG")) pa.maxseg = (uint32_t)std::max(0, atoi(ev_));   // tuning hook
        if (!lds_atomics) pa.dbg_flags |= 2u;
        if (ws.variant & W3_VAR_INJECT_LDS_FAULT) pa.dbg_flags |= 8u;
        pa.fault_block = ws.fault_block;
        bytes += n * 17;
        if (c == LEAF_SMALL_AC) {
            w3::HashArgs ha;
            memset(&ha, 0, sizeof ha);
            ha.in = d_in; ha.n = n; ha.block_size = (uint32_t)block_size; ha.max_bits = nd.max_bits;
            ha.hmask = (1u << (nd.bits - 3)) - 1u; ha.keys = (uint2 *)ws.keys;
            memcpy(ha.table, nd.table, sizeof ha.table);
            if ((rc = tp_ensure(ws.achash_lut, ws.achash_lut_cap, (size_t)(8u << W3_ACHASH_LUT_BITS) * 18, err))) return rc;
            ha.lut = (uint4 *)ws.achash_lut;
            ha.lut_key = (uint16_t *)((uint8_t *)ws.achash_lut + (size_t)(8u << W3_ACHASH_LUT_BITS) * 16);
            if (ev && !achash_timed) (void)hipEventRecord(ev[2 * W3_EV_ACHASH], s);   // (timed for the first ACHistory leaf)
            hipLaunchKernelGGL(w3::k_achash_lut, dim3((8u << W3_ACHASH_LUT_BITS) / 256), dim3(256), 0, s, ha);
            hipLaunchKernelGGL(w3::k_achash, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ha);
            if (ev && !achash_timed) { (void)hipEventRecord(ev[2 * W3_EV_ACHASH + 1], s); achash_timed = true; ws.achash_timed = true; }
            pa.keys = (const uint2 *)ws.keys;
            launch_small<true>(nd.bits - 3, dim3(grid_small), s, pa);
            bytes += n * 17;
        } else if (c == LEAF_SMALL_HUFF) {
            if (!ws.huff) { err = "HuffHistory tables not staged"; return W3_E_HIP; }
            w3::HuffKeyArgs hk;
            memset(&hk, 0, sizeof hk);
            hk.in = d_in; hk.n = n; hk.block_size = (uint32_t)block_size; hk.hmask = (1u << (nd.bits - 3)) - 1u;
            hk.tb = ws.huff + nd.reserved; hk.keys = (uint2 *)ws.keys;
            // blocks whose bounded walk-back met a run of zero-length codes are redone serially (k_huffkeys_fix)
            if ((rc = tp_ensure(ws.huff_redo, ws.huff_redo_cap, (size_t)nb * 4, err))) return rc;
            hk.redo = (uint32_t *)ws.huff_redo; hk.nblocks = nb;
            (void)hipMemsetAsync(ws.huff_redo, 0, (size_t)nb * 4, s);
            hipLaunchKernelGGL(w3::k_huffkeys<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, hk);
            hipLaunchKernelGGL(w3::k_huffkeys_fix<false>, dim3((nb + 63u) / 64u), dim3(64), 0, s, hk);
            pa.keys = (const uint2 *)ws.keys;
            launch_small<true>(nd.bits - 3, dim3(grid_small), s, pa);
            bytes += n * 17;
        } else if (c == LEAF_SMALL) {
            // beside a wide leaf's rank kernel (bound by its scattered stores) instead of beside the partition passes: see below
            // (Two encodes in flight — join_all: at once on the launch stream, beside the partition passes on the side stream: the first
            // predict half 15.4 -> 14.3 ms, 68.3 -> 67.6 ms per step; W3_OPT_TUNE bit 11 keeps the deferred order there too.)
            if (n_small_def < 8 && (!join_all || (ws.tune & 2048u))) small_def[n_small_def++] = pa;
            else {
                const bool timed = ev && !ws.small_timed;
                if (timed) { (void)hipEventRecord(ev[2 * W3_EV_SMALL], s); ws.small_timed = true; }
                launch_small_shaped(ws, nd.bits - 3, nb, s, pa);
                if (timed) (void)hipEventRecord(ev[2 * W3_EV_SMALL + 1], s);
            }
        } else {
            const int w = n_def;
            pa.perm = (uint32_t *)ws.perm_w[w];
            if (ws.debug_stamps) {
                if (!ws.dbg && hipMalloc(&ws.dbg, 64) != hipSuccess) ws.dbg = nullptr;
                if (ws.dbg) { (void)hipMemsetAsync(ws.dbg, 0, 64, s); pa.dbg = (unsigned long long *)ws.dbg; }
            }
            pa.rec = (uint2 *)ws.rec_w[w]; pa.splits = (uint32_t *)ws.splits_w[w];
            pa.job_counter = (uint32_t *)ws.splits_w[w] + (size_t)nb * (W3_SLICES + 1);   // lives behind the split table
            (void)hipMemsetAsync(pa.job_counter, 0, 4, sp);
            if (const char *ev_ = w3_tune_env("W3_DEBUG_NOSTORE")) pa.dbg_flags = (uint32_t)atoi(ev_) & 5u;   // -DW3_TUNING timing experiments: results are wrong
            if (!lds_atomics) pa.dbg_flags |= 2u;
            // A persistent grid of 2048 wavefronts walks the (block, slice) jobs in block order, so only the blocks in flight
            // (~32 with 64 slices each, more while a block's largest group is still running) are being scattered into and their
            // P regions stay in the Infinity Cache.  Measured at 1e9 B, whole predict phase (grid 1024 / 1536 / 2048 / 2560 / 3072 /
            // 4096): 64.0 / 59.1 / 54.6 / 55.1 / 58.7 / 58.7 ms.
            uint32_t rank_waves = 2048u;
            if (const char *ev_ = w3_tune_env("W3_RANK_GRID")) rank_waves = (uint32_t)std::max(64, atoi(ev_));   // tuning hook
            // half-CU shapes: the FIRST rank kernel starts beside the previous call's coder and must leave it its half of every CU's
            // LDS whichever of the two is dispatched first; the later ones are single wavefronts again (they fill what is free)
            const uint32_t nw_rank = (ws.tune & 32768u) ? W3_NW_RANK8 : W3_NW_RANK;
            const uint32_t grid_rank = (ws.half_cu && (w == 0 || (ws.tune & (32u | 64u))) && !(ws.tune & 8u)) ? std::min<uint32_t>((nb * W3_SLICES + nw_rank - 1) / nw_rank, W3_HALF_CU_GRID)
                                                              : std::min<uint32_t>(nb * W3_SLICES, rank_waves);
            // an order-2 leaf behind an Order1 leaf starts from that leaf's records (sorted by c1; same stream, so they are ready)
            const bool chained = c == LEAF_WIDE2 && ws.wide1_slot >= 0 && !(ws.variant & W3_VAR_NO_CHAINED_PARTITION);
            if (chained) pa.rec_src = (const uint2 *)ws.rec_w[ws.wide1_slot];
            // one 8-bit pass through LDS tiles (k_partition8) needs the lane-ordered LDS adds; otherwise 4-bit passes
            const bool p8 = lds_atomics && !(ws.variant & W3_VAR_PARTITION4);
            const uint32_t grid_p8h = std::min<uint32_t>((nb + W3_NW_PART - 1) / W3_NW_PART, 2 * W3_HALF_CU_GRID);
            const bool p8h = p8 && ws.half_cu && (ws.tune & 32u);   // W3_OPT_TUNE bit 5: the first predict half in half-CU shapes too
            if (ev && w < 4) (void)hipEventRecord(ev[2 * (W3_EV_PART0 + w)], sp);
            if (c == LEAF_WIDE1 && p8h) hipLaunchKernelGGL((w3::k_partition8<1, W3_NW_PART>), dim3(grid_p8h), dim3(64 * W3_NW_PART), 0, sp, pa);
            else if (chained && p8h) hipLaunchKernelGGL((w3::k_partition8<3, W3_NW_PART>), dim3(grid_p8h), dim3(64 * W3_NW_PART), 0, sp, pa);
            else if (c == LEAF_WIDE1 && p8) hipLaunchKernelGGL(w3::k_partition8<1>, dim3(grid_wide), dim3(64), 0, sp, pa);
            else if (chained && p8) hipLaunchKernelGGL(w3::k_partition8<3>, dim3(grid_wide), dim3(64), 0, sp, pa);
            else if (c == LEAF_WIDE1) hipLaunchKernelGGL(w3::k_partition<1>, dim3(grid_wide), dim3(64), 0, sp, pa);
            else if (chained) hipLaunchKernelGGL(w3::k_partition<3>, dim3(grid_wide), dim3(64), 0, sp, pa);
            else hipLaunchKernelGGL(w3::k_partition<2>, dim3(grid_wide), dim3(64), 0, sp, pa);
            if (ev && w < 4) (void)hipEventRecord(ev[2 * (W3_EV_PART0 + w) + 1], sp);
            if (forked) (void)hipEventRecord(ws.ev_join[n_def], ws.side);
            if (c == LEAF_WIDE1) ws.wide1_slot = n_def;
            deferred[n_def].pa = pa; deferred[n_def].cls = c; deferred[n_def].grid_rank = grid_rank; n_def++;
            bytes += n * 16 * (c == LEAF_WIDE1 || chained ? 2 : 4);  // record passes: 8 B written + 8 B read each
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = std::string("predict launch: ") + hipGetErrorString(e); return W3_E_HIP; }
    }
    // The time-ordered leaves (VALU-bound, coalesced stores) go to the side stream BEHIND the partition passes, so they run
    // beside the LAST rank kernel, which leaves VALU and most of the LDS free (2048 waves of 12 KiB); beside the first
    // partition pass — where they used to start — they only delayed it: k_partition8<1> took 14.4 ms instead of 7.2
    // (timeline, profiles/r2_final/timeline_before_reorder.txt) because k_predict_small's 5120 waves filled the CUs first.
    if (n_small_def) {
        // (Measured: with 3 .. 8 instead of 20 waves per CU, so that it fits beside the rank kernel's 8 from the start, the phase
        // takes the same 43-44 ms — the two kernels' times add up either way.)
        if (ev) { (void)hipEventRecord(ev[2 * W3_EV_SMALL], sp); ws.small_timed = true; }
        for (int k = 0; k < n_small_def; k++) launch_small_shaped(ws, (int)small_def[k].hbits, nb, sp, small_def[k]);
        if (ev) (void)hipEventRecord(ev[2 * W3_EV_SMALL + 1], sp);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = std::string("predict launch: ") + hipGetErrorString(e); return W3_E_HIP; }
        if (forked) (void)hipEventRecord(ws.ev_small, ws.side);
    }
    if (join_all && forked) {
        if (n_def) (void)hipStreamWaitEvent(s, ws.ev_join[n_def - 1], 0);
        if (n_small_def) (void)hipStreamWaitEvent(s, ws.ev_small, 0);
    }
    st.forked = forked; st.lds_atomics = lds_atomics; st.n_def = n_def; st.n_small_def = n_small_def; st.n_live = n_live;
    st.bytes = bytes; st.slot_stride = slot_stride;
    return W3_OK;
}

// need_P: also merge the leaves' streams into ws.P (k_mix).  The default coder mixes on the fly and needs no P for up to 4 live leaves.
static inline int twophase_predict_b(TwoPhaseWs &ws, hipStream_t s, const ParsedSpec &ps, const uint8_t *d_in, size_t n, size_t block_size,
                                     uint32_t nb, bool need_P, const uint16_t **d_P, hipEvent_t *ev, w3_timing *tm, std::string &err) {
    int rc = W3_OK;
    TwoPhaseWs::PredictState &st = ws.pst;
    auto &deferred = st.deferred;
    const bool forked = st.forked;
    const int n_def = st.n_def, n_small_def = st.n_small_def, n_live = st.n_live;
    uint64_t bytes = st.bytes;
    const uint64_t slot_stride = st.slot_stride;
    w3::SlotArgs &sa = st.sa;
    w3::MixArgs &ma = ws.mix;
    // join, then rank inside the sorted groups (main stream: these kernels want the Infinity Cache to themselves)
    // (measured, no gain: the later leaves' rank kernels on the side stream beside the first one — 35 + 28 ms together against
    // 17.7 + 15.8 ms one after the other: they saturate the same scattered-store path — and the time-ordered leaves last)
    // (The first rank kernel used to start as soon as ITS records were sorted, beside the next leaf's partition pass: two kernels
    // bound by the same non-coalesced store path — together 21.5 ms, one after the other 3.5 + 14: timeline_after_reorder.txt.)
    for (int w = 0; w < n_def; w++) {
        if (forked) (void)hipStreamWaitEvent(s, ws.ev_join[n_def - 1], 0);   // every leaf's records are sorted
        if (ev) (void)hipEventRecord(ev[2 * (W3_EV_RANK0 + w)], s);
        if (ws.half_cu && (w == 0 || (ws.tune & (32u | 64u))) && !(ws.tune & 8u) && (ws.tune & 32768u)) {
            if (deferred[w].cls == LEAF_WIDE1) hipLaunchKernelGGL((w3::k_rank_sorted<1, W3_NW_RANK8, 4>), dim3(deferred[w].grid_rank), dim3(64 * W3_NW_RANK8), 0, s, deferred[w].pa);
            else hipLaunchKernelGGL((w3::k_rank_sorted<2, W3_NW_RANK8, 4>), dim3(deferred[w].grid_rank), dim3(64 * W3_NW_RANK8), 0, s, deferred[w].pa);
        } else if (ws.half_cu && (ws.tune & 32768u)) {   // the later rank kernels: single wavefronts of 10 KiB (eight beside the coder's 74 KiB)
            if (deferred[w].cls == LEAF_WIDE1) hipLaunchKernelGGL((w3::k_rank_sorted<1, 1, 4>), dim3(deferred[w].grid_rank), dim3(64), 0, s, deferred[w].pa);
            else hipLaunchKernelGGL((w3::k_rank_sorted<2, 1, 4>), dim3(deferred[w].grid_rank), dim3(64), 0, s, deferred[w].pa);
        } else if (ws.half_cu && (w == 0 || (ws.tune & (32u | 64u))) && !(ws.tune & 8u)) {
            if (deferred[w].cls == LEAF_WIDE1) hipLaunchKernelGGL((w3::k_rank_sorted<1, W3_NW_RANK>), dim3(deferred[w].grid_rank), dim3(64 * W3_NW_RANK), 0, s, deferred[w].pa);
            else hipLaunchKernelGGL((w3::k_rank_sorted<2, W3_NW_RANK>), dim3(deferred[w].grid_rank), dim3(64 * W3_NW_RANK), 0, s, deferred[w].pa);
        } else if (deferred[w].cls == LEAF_WIDE1) hipLaunchKernelGGL(w3::k_rank_sorted<1>, dim3(deferred[w].grid_rank), dim3(64), 0, s, deferred[w].pa);
        else hipLaunchKernelGGL(w3::k_rank_sorted<2>, dim3(deferred[w].grid_rank), dim3(64), 0, s, deferred[w].pa);
        if (ev) (void)hipEventRecord(ev[2 * (W3_EV_RANK0 + w) + 1], s);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = std::string("rank launch: ") + hipGetErrorString(e); return W3_E_HIP; }
    }
    ws.n_wide = n_def;
    if (n_small_def && forked) (void)hipStreamWaitEvent(s, ws.ev_small, 0);
    // Slot-state leaves.  Few blocks: the sorted replay of w3_slot2.h (a wavefront per block; k_slot's lane per block costs the
    // lone-lane latency of a whole block, ~200 ms per 64 KiB of block, however few there are).  Many blocks (measured crossover
    // ~7,000: k_slot 375 ms against 460 at 15,259 blocks): k_slot.  The replay needs the lane-ordered LDS adds (its partition).
    bool slot_sorted = false;
    ws.slot_sorted = false;
    if (sa.n_leaves && !(ws.variant & W3_VAR_SLOT_TABLE) && st.lds_atomics && block_size <= (1ull << 31)) {
        slot_sorted = nb < W3_SLOT_SORTED_MAX_BLOCKS || (ws.variant & W3_VAR_SLOT_SORTED);
        bool two_passes = false;
        for (int l = 0; l < sa.n_leaves; l++) { slot_sorted &= sa.leaf[l].log_cells <= 16u; two_passes |= sa.leaf[l].log_cells > 8u; }
        const size_t key_bytes = (size_t)sa.n_leaves * 2 * n * 8, hist_bytes = (size_t)sa.n_leaves * nb * (512 + 1024) * 4;   // (digit counts + fine-bin prefixes)
        if (slot_sorted && ws.slot_keys_cap < 2 * key_bytes + hist_bytes + 64) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); err = "hipMemGetInfo failed"; return W3_E_HIP; }
            if (2 * key_bytes + hist_bytes + (8ull << 30) + n * 40 > free_b + ws.slot_keys_cap) slot_sorted = false;   // (the records would not fit: k_slot batches its tables)
        }
        if (slot_sorted) {
            if (!ws.st) { err = "state table not staged"; return W3_E_HIP; }
            if ((rc = tp_ensure(ws.slot_keys, ws.slot_keys_cap, 2 * key_bytes + hist_bytes + 64, err))) return rc;
            w3::Slot2Args s2;
            memset(&s2, 0, sizeof s2);
            s2.in = d_in; s2.n = n; s2.block_size = (uint32_t)block_size; s2.nblocks = nb;
            s2.keys_a = (uint64_t *)ws.slot_keys; s2.keys_b = (uint64_t *)((uint8_t *)ws.slot_keys + key_bytes);
            s2.hist = (uint32_t *)((uint8_t *)ws.slot_keys + 2 * key_bytes);
            s2.pre2 = s2.hist + (size_t)sa.n_leaves * nb * 512;
            if ((rc = tp_ensure(ws.dummy, ws.dummy_cap, 2048, err))) return rc;
            s2.st = ws.st; s2.n_leaves = sa.n_leaves; s2.fault = ws.order_fault; s2.dummy = (uint8_t *)ws.dummy;
            for (int l = 0; l < sa.n_leaves; l++) s2.leaf[l] = sa.leaf[l];
            if (ev) (void)hipEventRecord(ev[2 * W3_EV_SLOT], s);
            const dim3 gw(std::min<uint32_t>(nb, 256 * 16), sa.n_leaves);
            hipLaunchKernelGGL(w3::k_slot_events, gw, dim3(64), 0, s, s2);
            hipLaunchKernelGGL(w3::k_slot_sort<0>, gw, dim3(64), 0, s, s2);
            if (two_passes) hipLaunchKernelGGL(w3::k_slot_sort<1>, gw, dim3(64), 0, s, s2);   // (a leaf of at most 2^8 Cells passes through unchanged: one bin)
            // replay: jobs (block, leaf, wavefront) from one counter, block-major, to a persistent grid (one 8-wave workgroup per CU)
            s2.jobs_per_block = 0;
            for (int l = 0; l < sa.n_leaves; l++) {
                const uint32_t cells = 1u << sa.leaf[l].log_cells;
                // wavefronts per (block, leaf): 4 (four fine bins per lane).  Measured at enwik8 size, four 2^14-cell leaves (tools/r3_sl2.sh; replay kernel,
                // with / without its probability stores): 1 wavefront 36.3 / 13.7 ms (2,048 streams live: the 8-byte stores reach HBM as partial
                // writes), 4 wavefronts 25.8 / 22.4, 16 wavefronts 41.8 / 41.2 — there the stores merge in the Infinity Cache and cost nothing, but a
                // lane owns only 16 Cells and the wavefront runs as long as its busiest lane (hashed text contexts are very unevenly used: 6.6 us per
                // event round against 2.2 with 256 Cells per lane); whether the ranges come from a binary search or from this prefix table made no
                // difference.  W3_OPT_TUNE bits 9 / 10: 16 / 1.
                const uint32_t wcap = (ws.tune & 1024u) ? 1u : (ws.tune & 512u) ? 16u : 4u;
                s2.leaf_w[l] = std::min<uint32_t>(wcap, std::max<uint32_t>(1u, std::min<uint32_t>(cells, 1024u) / 64u));
                s2.jobs_per_block += s2.leaf_w[l];
            }
#ifdef W3_TUNING
            s2.dbg = (ws.tune >> 8) & 1u;   // (-DW3_TUNING builds only — W3_OPT_TUNE bit 8: timing experiment, results WRONG)
#endif
            s2.job_counter = (uint32_t *)((uint8_t *)ws.slot_keys + 2 * key_bytes + hist_bytes);
            (void)hipMemsetAsync(s2.job_counter, 0, 4, s);
            const uint32_t njobs = nb * s2.jobs_per_block;
            hipLaunchKernelGGL(w3::k_slot_replay, dim3(std::min<uint32_t>((njobs + W3_S2_WAVES - 1) / W3_S2_WAVES, 256u)), dim3(64 * W3_S2_WAVES), 0, s, s2, two_passes ? 1 : 0);
            if (ev) (void)hipEventRecord(ev[2 * W3_EV_SLOT + 1], s);
            if (tm) tm->n_slot_launches = 0;   // (no table batches)
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { err = std::string("slot predict launch: ") + hipGetErrorString(e); return W3_E_HIP; }
            // per input byte and leaf: 2 events x (8 B record written, 1 or 2 passes of 8 r + 8 w, 8 B read, 8 B written) + the input
            bytes += (uint64_t)sa.n_leaves * n * (1 + 2 * (8 + (two_passes ? 32 : 16) + 8 + 8));
            ws.slot_sorted = true;
        }
    }
    if (sa.n_leaves && !slot_sorted) {
        // HashMaps in HBM, one per (block, leaf), zeroed per batch of blocks; as many blocks at once as the budget allows
        if (!ws.st) { err = "state table not staged"; return W3_E_HIP; }
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); err = "hipMemGetInfo failed"; return W3_E_HIP; }
        // budget = what is free now minus what this call still has to allocate after the slot leaves (the merged / APM
        // stream P, the records and scratch of an ORDER1 APM stage) and 4 GiB of slack for the caller
        uint64_t later = 4ull << 30;
        if (ws.P_cap < n * 16) later += n * 16;
        bool apm1 = false;
        for (int k = 0; k < ps.n_apm; k++) apm1 |= ps.apm[k].align == W3_APM_ORDER1;
        if (apm1) {
            if (ws.rec_cap < n * 8) later += n * 8;
            const uint64_t perm_need = (uint64_t)std::min<uint32_t>(nb, 256 * 16) * 2 * block_size * 8;
            if (ws.perm_cap < perm_need) later += perm_need;
        }
        const uint64_t avail = (uint64_t)free_b + ws.slot_tables_cap;
        uint64_t budget = avail > later ? avail - later : 0;
        if (ws.slot_budget_mb) budget = (uint64_t)ws.slot_budget_mb << 20;   // W3_OPT_SLOT_BUDGET_MB (tests: several batches on a small input)
        uint64_t lanes = std::min<uint64_t>(budget / slot_stride, nb);
        if (lanes < nb) {
            // equal batches: a batch costs at least the lone-wave latency of a whole block, however few lanes it has
            lanes = lanes / 64 * 64;
            if (lanes) { const uint64_t nbatch = (nb + lanes - 1) / lanes; lanes = std::min<uint64_t>(lanes, ((nb + nbatch - 1) / nbatch + 63) / 64 * 64); }
        }
        if (lanes == 0) { err = "slot-leaf hash maps of one wavefront (" + std::to_string(slot_stride * 64) + " B) exceed the device budget"; return W3_E_NOMEM; }
        if ((rc = tp_ensure(ws.slot_tables, ws.slot_tables_cap, (size_t)(lanes * slot_stride), err))) return rc;
        sa.in = d_in; sa.n = n; sa.block_size = (uint32_t)block_size; sa.tables = (uint8_t *)ws.slot_tables; sa.lane_stride = slot_stride;
        sa.st = ws.st;
        if ((rc = tp_ensure(ws.dummy, ws.dummy_cap, 2048, err))) return rc;
        sa.dummy = (uint8_t *)ws.dummy;
        if (const char *ev_ = w3_tune_env("W3_SLOT_DEBUG")) sa.dbg_flags = (uint32_t)atoi(ev_);   // -DW3_TUNING timing experiments only: results are wrong
        if (ev) (void)hipEventRecord(ev[2 * W3_EV_SLOT], s);
        for (uint32_t first = 0; first < nb; first += (uint32_t)lanes) {
            const uint32_t cnt = std::min<uint32_t>((uint32_t)lanes, nb - first);
            sa.first_block = first; sa.n_lanes = cnt;
            (void)hipMemsetAsync(ws.slot_tables, 0, (size_t)cnt * slot_stride, s);
            hipLaunchKernelGGL(w3::k_slot, dim3((cnt + 63) / 64, sa.n_leaves), dim3(64), 0, s, sa);
            if (tm) tm->n_slot_launches++;
        }
        if (ev) (void)hipEventRecord(ev[2 * W3_EV_SLOT + 1], s);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = std::string("slot predict launch: ") + hipGetErrorString(e); return W3_E_HIP; }
        bytes += (uint64_t)sa.n_leaves * n * (1 + 16 + 2 * 192);   // 2 nibbles x (96 B read + 96 B written) per input byte
    }
    if (n_live >= 1) ma.n_src = n_live;
    if (n_live == 1) ws.P_valid = true;   // the single leaf wrote ws.P itself
    if (need_P && !ws.P_valid) {
        if ((rc = twophase_mix(ws, s, n, err))) return rc;
        bytes += n * 16 * (n_live + 1);
    }
    if (ev) (void)hipEventRecord(ev[2 * W3_EV_PREDICT + 1], s);
    if (tm) tm->predict_bytes = bytes;
    if (d_P) *d_P = (const uint16_t *)ws.P;
    return W3_OK;
}

static inline int twophase_predict(TwoPhaseWs &ws, hipStream_t s, const ParsedSpec &ps, const uint8_t *d_in, size_t n, size_t block_size,
                                   uint32_t nb, bool need_P, const uint16_t **d_P, hipEvent_t *ev, w3_timing *tm, std::string &err) {
    int rc = twophase_predict_a(ws, s, ps, d_in, n, block_size, nb, ev, err, false);
    if (rc) return rc;
    return twophase_predict_b(ws, s, ps, d_in, n, block_size, nb, need_P, d_P, ev, tm, err);
}

#define W3_VERIFY_BLOCKS 16u   // sampled blocks per call, at least; max(this, nblocks / 256) are taken, and the sample ROTATES from call to call
                               // (a systematic change of hardware behaviour shows in any block; a fault confined to one block is
                               // met after at most nblocks / sample calls)

// Always-on insurance for the one undocumented hardware property the default predict kernels rely on (returning LDS adds of
// one wavefront resolve in ascending lane order: atomic_round, k_partition8).  After a predict phase that used it, up to W3_VERIFY_BLOCKS
// evenly spaced full-length blocks (at most 64 MiB) are predicted AGAIN with the ballot rounds and 4-bit partitions — exact
// by construction, no lane-order assumption — and every leaf's stream is compared on the device.  The main phase ran
// under the production load (all CUs, 8 waves per CU); that is the condition the per-context self-test cannot reproduce.
// d_mismatch (device word, zeroed by the caller) counts differing waves; the caller reads it with its status flags
// and, when it is not zero, re-encodes the whole call on the ballot path and keeps the context there.
static inline int twophase_verify(TwoPhaseWs &ws, hipStream_t s, const ParsedSpec &ps, const uint8_t *d_in, size_t n, size_t block_size,
                                  uint32_t nb, uint32_t *d_mismatch, std::string &err) {
    if (!ws.verify || !ws.used_lds_atomics) return W3_OK;
    const uint32_t nb_full = (uint32_t)(n / block_size);
    // ws.verify = v >= 1: v / 256 of the blocks (W3_OPT_VERIFY; 1 = the default sample), at most 64 MiB x v of input
    const uint64_t vv = (uint64_t)std::max(1, ws.verify);
    uint32_t S = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(nb_full, std::max<uint64_t>(W3_VERIFY_BLOCKS, nb_full * vv / 256u)), std::max<uint64_t>(1, vv * (64ull << 20) / block_size));
    size_t vn = (size_t)S * block_size;
    const uint32_t gap = S ? nb_full / S : 0u, rot = gap ? ws.verify_calls++ % gap : 0u;
    const uint8_t *vsrc = nullptr;
    int rc;
    if (nb_full == 0) { S = 1; vn = n; vsrc = d_in; (void)nb; }   // a single short block: verify it whole, in place
    else {
        if ((rc = tp_ensure(ws.vin, ws.vin_cap, vn, err))) return rc;
        hipLaunchKernelGGL(w3::k_gather_blocks, dim3(std::min<uint32_t>((uint32_t)((block_size + 255) / 256), 64u), S), dim3(256), 0, s,
                           d_in, (uint32_t)block_size, nb_full, S, rot, (uint8_t *)ws.vin);
        vsrc = (const uint8_t *)ws.vin;
    }
    if (!ws.vws) ws.vws = new TwoPhaseWs();
    TwoPhaseWs &v = *ws.vws;
    // (no side stream of its own: the sample is small, and every stream a context creates is one more claim on the hardware queues)
    v.variant = (ws.variant | W3_VAR_NO_LDS_ATOMICS | W3_VAR_PARTITION4 | W3_VAR_NO_SIDE_STREAM) & ~(uint32_t)W3_VAR_INJECT_LDS_FAULT;
    v.lds_order = 0; v.verify = 0;
    v.stretch = ws.stretch; v.squash = ws.squash; v.st = ws.st; v.huff = ws.huff; v.slot_budget_mb = ws.slot_budget_mb;
    // only the leaves whose kernels use the property are re-predicted (not the slot-state leaves, not k_predict_wave's)
    ParsedSpec vps;
    int map[W3_MAX_LEAVES], nmap = 0;   // vps leaf -> index among ws.mix.src (the live leaves of ps, in order)
    int live_idx = 0;
    for (int l = 0; l < ps.n_leaves; l++) {
        const int c = leaf_class(ps.leaf[l]);
        if (c == LEAF_FROZEN) continue;
        if (c != LEAF_SLOT && c != LEAF_WAVE) { vps.leaf[vps.n_leaves++] = ps.leaf[l]; map[nmap++] = live_idx; }
        live_idx++;
    }
    vps.n_huff = ps.n_huff; vps.huff = ps.huff;
    if (vps.n_leaves == 0) return W3_OK;
    if ((rc = twophase_predict(v, s, vps, vsrc, vn, block_size, (uint32_t)((vn + block_size - 1) / block_size), false, nullptr, nullptr, nullptr, err))) return rc;
    const uint32_t cmp_bs = nb_full ? (uint32_t)block_size : (uint32_t)n;
    for (int k = 0; k < nmap; k++)
        hipLaunchKernelGGL(w3::k_compare_blocks, dim3(std::min<uint32_t>((cmp_bs + 255u) / 256u, 64u), S), dim3(256), 0, s,
                           ws.mix.src[map[k]], v.mix.src[k], cmp_bs, nb_full ? nb_full : 1u, S, rot, d_mismatch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("verify launch: ") + hipGetErrorString(e); return W3_E_HIP; }
    return W3_OK;
}

// APM chain at the root (w3_apm.h): turns the leaves' streams into the final stream ws.P, stage by stage.
static inline int twophase_apm(TwoPhaseWs &ws, hipStream_t s, const ParsedSpec &ps, const uint8_t *d_in, size_t n, size_t block_size,
                               uint32_t nb, hipEvent_t *ev, w3_timing *tm, std::string &err) {
    if (ps.n_apm == 0) return W3_OK;
    if (!ws.stretch || !ws.squash) { err = "APM LUTs not staged"; return W3_E_HIP; }
    int rc = tp_ensure(ws.P, ws.P_cap, n * 16, err);
    if (rc) return rc;
    if ((rc = tp_ensure(ws.dummy, ws.dummy_cap, 2048, err))) return rc;
    if (ev) (void)hipEventRecord(ev[2 * W3_EV_APM], s);
    uint64_t bytes = 0;
    bool partitioned = false;
    for (int k = 0; k < ps.n_apm; k++) {
        w3::ApmArgs aa;
        memset(&aa, 0, sizeof aa);
        aa.in = d_in; aa.n = n; aa.block_size = (uint32_t)block_size; aa.nblocks = nb;
        aa.P = (uint16_t *)ws.P; aa.stretch = ws.stretch; aa.squash = ws.squash; aa.rate = ps.apm[k].max_bits;
        aa.dummy = (uint16_t *)ws.dummy;
        aa.oob = ws.apm_oob;
        if (ps.apm[k].align == W3_APM_ORDER0) {
            int L = 1;
            if (ws.P_valid) aa.src[0] = (const uint16_t *)ws.P;
            else if (ws.mix.n_src <= 8) { L = ws.mix.n_src; for (int l = 0; l < L; l++) aa.src[l] = (const uint16_t *)ws.mix.src[l]; }
            else { if ((rc = twophase_mix(ws, s, n, err))) return rc; aa.src[0] = (const uint16_t *)ws.P; bytes += n * 16 * (ws.mix.n_src + 1); }
            const dim3 grid((nb + W3_APM_WAVES - 1) / W3_APM_WAVES), blk(128 * W3_APM_WAVES);   // two wavefronts per block
            // W3_OPT_TUNE bit 1: dynamic LDS on top of the kernel's 79,968 B, so that two of its workgroups do not fit a CU
            const uint32_t dyn = (ws.half_cu && (ws.tune & 2u)) ? 2048u : 0u;
#define W3_LAUNCH_APM0(LL)                                                                                                   \
            do {                                                                                                             \
                if (dyn) (void)hipFuncSetAttribute((const void *)w3::k_apm0<LL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
                hipLaunchKernelGGL(w3::k_apm0<LL>, grid, blk, dyn, s, aa);                                                   \
            } while (0)
            switch (L) {
            case 1: W3_LAUNCH_APM0(1); break;
            case 2: W3_LAUNCH_APM0(2); break;
            case 3: W3_LAUNCH_APM0(3); break;
            case 4: W3_LAUNCH_APM0(4); break;
            case 5: W3_LAUNCH_APM0(5); break;
            case 6: W3_LAUNCH_APM0(6); break;
            case 7: W3_LAUNCH_APM0(7); break;
            default: W3_LAUNCH_APM0(8); break;
            }
#undef W3_LAUNCH_APM0
            bytes += n * (16 * (uint64_t)L + 1 + 16);
        } else {
            if (!ws.P_valid) { if ((rc = twophase_mix(ws, s, n, err))) return rc; bytes += n * 16 * (ws.mix.n_src + 1); }
            // the records sorted by the previous byte: an Order1-shaped leaf of this call has already produced them
            const bool reuse = ws.wide1_slot >= 0;
            if (!reuse && (rc = tp_ensure(ws.splits, ws.splits_cap, (size_t)nb * (W3_SLICES + 1) * 4 + 64, err))) return rc;
            void *rec_p = reuse ? ws.rec_w[ws.wide1_slot] : nullptr, *splits_p = reuse ? ws.splits_w[ws.wide1_slot] : ws.splits;
            uint32_t *job_counter = (uint32_t *)splits_p + (size_t)nb * (W3_SLICES + 1);
            if (!partitioned && !reuse) {   // records sorted by the previous byte (the leaves' own partitions may have reused ws.rec since)
                const uint32_t grid_wide = std::min<uint32_t>(nb, 256 * 16);
                if ((rc = tp_ensure(ws.perm, ws.perm_cap, (size_t)grid_wide * 2 * block_size * 8, err))) return rc;
                if ((rc = tp_ensure(ws.rec, ws.rec_cap, n * 8, err))) return rc;
                w3::PredictArgs pa;
                memset(&pa, 0, sizeof pa);
                pa.in = d_in; pa.n = n; pa.block_size = (uint32_t)block_size; pa.nblocks = nb;
                pa.perm = (uint32_t *)ws.perm; pa.rec = (uint2 *)ws.rec; pa.splits = (uint32_t *)ws.splits; pa.job_counter = job_counter;
                hipLaunchKernelGGL(w3::k_partition<1>, dim3(grid_wide), dim3(64), 0, s, pa);
                partitioned = true;
                bytes += n * (4 + 32);
            }
            (void)hipMemsetAsync(job_counter, 0, 4, s);
            aa.rec = (const uint2 *)(reuse ? rec_p : ws.rec); aa.splits = (const uint32_t *)splits_p; aa.job_counter = job_counter;
            const uint32_t grid = std::min<uint32_t>((nb * W3_SLICES + W3_APM_WAVES - 1) / W3_APM_WAVES, 512u);   // 2 workgroups per CU: ~128 blocks live
            hipLaunchKernelGGL(w3::k_apm1, dim3(grid), dim3(64 * W3_APM_WAVES), 0, s, aa);
            bytes += n * (8 + 16 + 16);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = std::string("apm launch: ") + hipGetErrorString(e); return W3_E_HIP; }
        ws.P_valid = true;
    }
    memset(&ws.mix, 0, sizeof ws.mix);
    ws.mix.src[0] = (const uint4 *)ws.P; ws.mix.n_src = 1; ws.mix.P = (uint4 *)ws.P; ws.mix.n = n;
    if (ev) (void)hipEventRecord(ev[2 * W3_EV_APM + 1], s);
    if (tm) tm->predict_bytes += bytes;
    return W3_OK;
}

// The encode in three pieces, so that w3_encode_submit can interleave two calls:
//   twophase_predict (a + b)   on the predict stream
//   tp_after_predict           verification of a single-leaf APM spec in place; records "predict phase through"
//   tp_code_stage              APM stages, sampled verification (forked), coder — on the code stream
// twophase_encode below is the three in sequence (one stream for both = one call at a time, phases strictly in order).
struct TpPlan { int n_live, coder; bool x3, need_P, verify_on, verify_in_place; };
static inline TpPlan tp_plan(const TwoPhaseWs &ws, const ParsedSpec &ps) {
    TpPlan p;
    p.n_live = 0;
    for (int l = 0; l < ps.n_leaves; l++) p.n_live += leaf_class(ps.leaf[l]) != LEAF_FROZEN;
    p.coder = ws.half_cu && ws.coder_mode == 0 ? 5 : ws.coder_mode;   // half-CU shapes: k_coder_x5 in k_coder_x4's place
    p.x3 = (p.coder == 0 || p.coder == 4 || p.coder == 5) && (p.n_live <= 4 || ps.n_apm > 0);   // more leaves: merge with k_mix first, then k_coder_x2
    p.need_P = !p.x3 && ps.n_apm == 0;
    p.verify_on = false; p.verify_in_place = false;
    return p;
}

static inline int tp_after_predict(TwoPhaseWs &ws, hipStream_t s_pred, const ParsedSpec &ps, const uint8_t *d_in, size_t n, size_t block_size, uint32_t nb,
                                   uint32_t *d_flag, std::string &err) {
    const TpPlan pl = tp_plan(ws, ps);
    // flag word 2: mismatching waves.  The re-prediction only reads the leaves' streams, so it runs on its own stream (2.8 ms of
    // small launches otherwise) beside the CODER kernel, which leaves the chip's memory system and most of its issue slots
    // idle (beside k_apm0 its workgroups displaced some of that kernel's for ~1.4 ms per step, measured) — unless an APM stage
    // is about to rewrite the single leaf's stream in place: then it runs first, on the predict stream.
    const bool verify_on = ws.verify && ws.used_lds_atomics;
    const bool verify_in_place = verify_on && pl.n_live == 1 && ps.n_apm > 0;
    int rc;
    if (verify_in_place && (rc = twophase_verify(ws, s_pred, ps, d_in, n, block_size, nb, d_flag + 2, err))) return rc;
    if (!ws.ev_pred_done && hipEventCreateWithFlags(&ws.ev_pred_done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); err = "event creation failed"; return W3_E_HIP; }
    (void)hipEventRecord(ws.ev_pred_done, s_pred);
    return W3_OK;
}

// wait_ev: one more event the stage waits for (the NEXT call's first predict half); rec_after_apm: recorded when the APM stages are
// through (the next call's rank kernels wait for it).  Either may be null.
static inline int tp_code_stage(TwoPhaseWs &ws, hipStream_t s_pred, hipStream_t s, const ParsedSpec &ps, const uint8_t *d_in, size_t n, size_t block_size,
                                uint32_t nb, uint8_t *stripes, uint32_t stripe_cap, uint32_t *d_lens, uint32_t *d_flag, hipEvent_t wait_ev, hipEvent_t rec_after_apm,
                                hipEvent_t *ev, w3_timing *tm, std::string &err) {
    const TpPlan pl = tp_plan(ws, ps);
    const int n_live = pl.n_live, coder = pl.coder;
    const bool x3 = pl.x3;
    int rc;
    bool verify_forked = false;
    const bool verify_on = ws.verify && ws.used_lds_atomics;
    const bool verify_in_place = verify_on && n_live == 1 && ps.n_apm > 0;
    if (verify_on && !verify_in_place && (!ws.vstream || !ws.ev_v0)) {
        // (lowest priority level: its own hardware queues, apart from the launch streams'; and the re-prediction is in nobody's way)
        int lo_p = 0, hi_p = 0;
        bool ok = hipDeviceGetStreamPriorityRange(&lo_p, &hi_p) == hipSuccess && (ws.vstream || hipStreamCreateWithPriority(&ws.vstream, hipStreamNonBlocking, lo_p) == hipSuccess) &&
                  hipEventCreateWithFlags(&ws.ev_v0, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ws.ev_v1, hipEventDisableTiming) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); err = "verification stream creation failed"; return W3_E_HIP; }
    }
    if (s != s_pred) (void)hipStreamWaitEvent(s, ws.ev_pred_done, 0);
    if (wait_ev) (void)hipStreamWaitEvent(s, wait_ev, 0);
    // join on every way out of this function: the caller reads the mismatch word (and may free buffers) next
    struct Join { hipStream_t st; hipEvent_t ev; const bool &on; ~Join() { if (on) (void)hipStreamWaitEvent(st, ev, 0); } } join{s, ws.ev_v1, verify_forked};
    const auto leaf_streams = ws.mix;   // (the APM stages below replace ws.mix by their one output stream)
    auto fork_verify = [&](hipStream_t from) -> int {
        (void)hipEventRecord(ws.ev_v0, from); (void)hipStreamWaitEvent(ws.vstream, ws.ev_v0, 0);
        const auto after_apm = ws.mix;
        ws.mix = leaf_streams;
        const int vr = twophase_verify(ws, ws.vstream, ps, d_in, n, block_size, nb, d_flag + 2, err);
        ws.mix = after_apm;
        (void)hipEventRecord(ws.ev_v1, ws.vstream);
        verify_forked = true;
        return vr;
    };
    if ((rc = twophase_apm(ws, s, ps, d_in, n, block_size, nb, ev, tm, err))) return rc;   // leaves ws.P as the one source stream
    if (rec_after_apm) (void)hipEventRecord(rec_after_apm, s);
    if (verify_on && !verify_in_place && (rc = fork_verify(s))) return rc;
    if (w3_tune_env("W3_DEBUG_NOSTORE")) { err = "W3_DEBUG_NOSTORE: predict-only timing experiment"; return W3_E_UNSUPPORTED; }
    if ((rc = tp_ensure(ws.redo, ws.redo_cap, (size_t)nb * 4, err))) return rc;
    const uint32_t limit = std::min<uint32_t>(ws.acc_limit, 46u);
    if (ev) (void)hipEventRecord(ev[2 * W3_EV_CODER], s);
    if (x3) {
        w3::Coder3Args c3;
        memset(&c3, 0, sizeof c3);
        c3.in = d_in; c3.n = n; c3.block_size = (uint32_t)block_size; c3.nblocks = nb;
        for (int l = 0; l < ws.mix.n_src; l++) c3.src[l] = ws.mix.src[l];
        c3.stripes = stripes; c3.stripe_cap = stripe_cap; c3.out_len = d_lens; c3.flags = d_flag; c3.redo = (uint32_t *)ws.redo;
        c3.acc_limit = limit;
        c3.out_bits = ws.out_bits;
        c3.prio_mo = (ws.tune >> 7) & 1u;
        const dim3 grid((nb + 63) / 64), blk(192);
        if (coder == 4) {
            switch (ws.mix.n_src) {
            case 1: hipLaunchKernelGGL(w3::k_coder_x3<1>, grid, blk, 0, s, c3); break;
            case 2: hipLaunchKernelGGL(w3::k_coder_x3<2>, grid, blk, 0, s, c3); break;
            case 3: hipLaunchKernelGGL(w3::k_coder_x3<3>, grid, blk, 0, s, c3); break;
            default: hipLaunchKernelGGL(w3::k_coder_x3<4>, grid, blk, 0, s, c3); break;
            }
        } else if (coder == 5) {   // half the LDS: leaves room beside it for the next call's predict workgroups
            const dim3 blk2(256);
            switch (ws.mix.n_src) {
            case 1: hipLaunchKernelGGL(w3::k_coder_x5<1>, grid, blk, 0, s, c3); break;
            case 2: hipLaunchKernelGGL(w3::k_coder_x5<2>, grid, blk2, 0, s, c3); break;
            case 3: hipLaunchKernelGGL(w3::k_coder_x5<3>, grid, blk2, 0, s, c3); break;
            default: hipLaunchKernelGGL(w3::k_coder_x5<4>, grid, blk2, 0, s, c3); break;
            }
        } else {
            const dim3 blk2(256);   // two M-waves when the leaves are mixed on the fly
            switch (ws.mix.n_src) {
            case 1: hipLaunchKernelGGL(w3::k_coder_x4<1>, grid, blk, 0, s, c3); break;
            case 2: hipLaunchKernelGGL(w3::k_coder_x4<2>, grid, blk2, 0, s, c3); break;
            case 3: hipLaunchKernelGGL(w3::k_coder_x4<3>, grid, blk2, 0, s, c3); break;
            default: hipLaunchKernelGGL(w3::k_coder_x4<4>, grid, blk2, 0, s, c3); break;
            }
        }
        if (tm) tm->coder_bytes = (uint64_t)n * (16 * ws.mix.n_src + 1);
    } else {
        w3::CoderArgs ca;
        memset(&ca, 0, sizeof ca);
        ca.in = d_in; ca.n = n; ca.block_size = (uint32_t)block_size; ca.nblocks = nb; ca.P = (const uint4 *)ws.P;
        ca.stripes = stripes; ca.stripe_cap = stripe_cap; ca.out_len = d_lens; ca.flags = d_flag;
        ca.acc_limit = limit;
        ca.out_bits = ws.out_bits;
        if (coder == 2) {
            ca.redo = nullptr;
            hipLaunchKernelGGL(w3::k_coder, dim3((nb + 63) / 64), dim3(64), 0, s, ca);
        } else {
            ca.redo = (uint32_t *)ws.redo;
            if (coder == 1) hipLaunchKernelGGL(w3::k_coder_fast, dim3((nb + 63) / 64), dim3(64), 0, s, ca);
            else hipLaunchKernelGGL(w3::k_coder_x2, dim3((nb + 63) / 64), dim3(128), 0, s, ca);
        }
        if (tm) tm->coder_bytes = (uint64_t)n * 17;
    }
    if (ev) (void)hipEventRecord(ev[2 * W3_EV_CODER + 1], s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("coder launch: ") + hipGetErrorString(e); return W3_E_HIP; }
    if (tm) tm->n_coder_launches = 1;
    return W3_OK;
}

// s_pred: the predict phase's launch stream; s: the stream of the APM stages, the coder and whatever the caller enqueues next
// (pack, status read-back).  One stream for both = one call at a time, phases strictly in sequence.
static inline int twophase_encode(TwoPhaseWs &ws, hipStream_t s_pred, hipStream_t s, const ParsedSpec &ps, const uint8_t *d_in, size_t n, size_t block_size,
                                  uint32_t nb, uint8_t *stripes, uint32_t stripe_cap, uint32_t *d_lens, uint32_t *d_flag, hipEvent_t *ev,
                                  w3_timing *tm, std::string &err) {
    int rc = twophase_predict(ws, s_pred, ps, d_in, n, block_size, nb, tp_plan(ws, ps).need_P, nullptr, ev, tm, err);
    if (rc) return rc;
    if ((rc = tp_after_predict(ws, s_pred, ps, d_in, n, block_size, nb, d_flag, err))) return rc;
    return tp_code_stage(ws, s_pred, s, ps, d_in, n, block_size, nb, stripes, stripe_cap, d_lens, d_flag, nullptr, nullptr, ev, tm, err);
}

// Blocks the fast coder gave up on (pending run longer than its accumulator): re-code with k_coder.
static inline int twophase_recode(TwoPhaseWs &ws, hipStream_t s, const uint8_t *d_in, size_t n, size_t block_size, uint32_t nb,
                                  uint8_t *stripes, uint32_t stripe_cap, uint32_t *d_lens, uint32_t *d_flag, uint32_t n_redo, std::string &err) {
    int rc = twophase_mix(ws, s, n, err);   // k_coder_x3 mixes on the fly; the robust coder wants the merged stream
    if (rc) return rc;
    w3::CoderArgs ca;
    memset(&ca, 0, sizeof ca);
    ca.in = d_in; ca.n = n; ca.block_size = (uint32_t)block_size; ca.nblocks = nb; ca.P = (const uint4 *)ws.P;
    ca.stripes = stripes; ca.stripe_cap = stripe_cap; ca.out_len = d_lens; ca.flags = d_flag;
    ca.redo = (uint32_t *)ws.redo; ca.n_redo = n_redo;
    ca.out_bits = ws.out_bits;
    hipLaunchKernelGGL(w3::k_coder, dim3((n_redo + 63) / 64), dim3(64), 0, s, ca);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("recode launch: ") + hipGetErrorString(e); return W3_E_HIP; }
    return W3_OK;
}
