// w3hip.hip — host side of the C ABI declared in include/w3hip.h, plus the
// kernel launches.  MI355X (gfx950) only.  No CPU fallback anywhere in this
// file: without a device every entry point fails with W3_E_HIP.
#include "../../include/w3hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "w3_spec.h"
#include "w3_huff.h"
#include "w3_generic.h"
#include "w3_cm.h"
#include "w3_decode_spec.h"
#include "w3_pack.h"
#include "w3_twophase.h"
#include "w3_selftest.h"
#include "w3_sweep.h"
#include "w3_rccl.h"

using namespace w3;

static_assert(sizeof(w3_node) == 24 && sizeof(w3_huff_table) == 1536 && sizeof(w3_model_spec) == 760, "ABI struct layout (tests/test_host_abi.py)");

// ---------------------------------------------------------------------------
// ctx
// ---------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// One encode in flight: its own workspace, so that call k+1's predict phase can run beside call k's APM and coder kernels
// (w3_encode_submit / w3_encode_wait).  Job 0's members live in w3_ctx itself (every synchronous call uses them).
struct JobState {
    int state = 0;                 // 0 idle, 1 enqueued (w3_encode_wait completes it), 2 ran synchronously inside w3_encode_submit
    hipEvent_t ev_done = nullptr, ev_in = nullptr;
    hipEvent_t ev_a = nullptr, ev_apm = nullptr;   // first predict half through / APM stages through (what the other job's kernels wait for)
    bool code_pending = false;     // the APM + coder + pack stage is not enqueued yet (it goes behind the NEXT job's first predict half)
    ParsedSpec ps; uint32_t nb = 0, cap = 0;
    uint32_t *h_status = nullptr;  // pinned: [0..3] the coder's flag words, [4..5] total compressed bytes
    // the submitted call, kept for the rare synchronous redo in w3_encode_wait (stripe overflow, fast-coder hand-back, LDS-order fault)
    w3_model_spec spec{};
    w3_huff_table huff_copy[W3_MAX_HUFF];
    const uint8_t *d_in = nullptr; size_t n = 0, block_size = 0; uint8_t *d_out = nullptr; size_t out_cap = 0;
    uint32_t *d_block_lens = nullptr; uint64_t *d_total = nullptr;
    w3_timing tm{};
    w3_timing tm_ev{}; bool tm_snap = false;   // the event times, collected early (a synchronous fallback is about to reuse job 0's events)
    int sync_rc = W3_OK;           // state 2: what the synchronous run inside w3_encode_submit returned
    uint64_t total_out = 0; bool total_valid = false;   // the compressed size w3_encode_wait read from the job's pinned status words
    bool has_apm = false, has_slot = false, timed = false;
    hipStream_t sc = nullptr;      // the stream this job's code stage runs on (free-running jobs: one each; ordered jobs share s_code[0])
};

// Jobs in flight: free-running — every code stage on its own stream, the calls' coders overlapping one another — while a call's
// coder leaves a good part of the chip idle: four up to W3_FREE_RUN4_BLOCKS blocks (64 coder workgroups), three up to
// W3_FREE_RUN_BLOCKS (measured: 7,629 blocks 36.6 ms per step with three, 39.1 with four, 43.8 with the ordered pair; 11,444 blocks
// 53.6 against 56.0 ordered; at 15,259 the two forms meet at 68-70 ms); beyond that the ordered pair of DESIGN.md section 2.8
// (a job workspace is ~70 bytes per input byte).
#define W3_MAX_JOBS 4
#define W3_FREE_RUN_BLOCKS 12288u
#define W3_FREE_RUN4_BLOCKS 4096u

// One host-buffer encode in flight (w3_encode_host_submit / w3_encode_host_wait): its own device input / output buffers, so that call
// k+1's input can travel over PCIe while call k is being encoded and call k-1's streams travel back.  One more than the device jobs:
// the extra one is the call whose input is on its way while every device job slot is busy.
#define W3_MAX_HOST_JOBS (W3_MAX_JOBS + 1)
struct HostJob {
    int state = 0;                 // 0 idle, 1 input enqueued (H2D), encode not submitted yet (no device job slot free), 2 encode submitted (djob),
                                   // 3 through on the device (rc / total known), output not fetched yet
    int djob = -1, rc = W3_OK;
    uint64_t seq = 0, total = 0;
    DevBuf d_in, d_out, d_lens, d_total;
    hipEvent_t ev_d2h = nullptr;
    w3_model_spec spec{};
    w3_huff_table huff_copy[W3_MAX_HUFF];
    size_t n = 0, block_size = 0, nb = 0, dcap = 0;
    uint8_t *out = nullptr; size_t out_cap = 0; uint32_t *block_lens = nullptr;
    w3_timing tm{};
};

struct w3_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int opt_path = W3_PATH_AUTO;
    int opt_timing = 0;
    w3_timing timing{};
    hipEvent_t ev[W3_NEV]{};
    // workspace
    DevBuf tables, stripes, lens, offs, total, flag, io_in, io_out, coffs, misc, cm_luts, achash_luts, huff, bits, sweep;
    TwoPhaseWs tp;
    // the second job of the submit / wait pipeline
    struct JobWs { TwoPhaseWs tp; DevBuf stripes, flag, bits, offs, huff; hipEvent_t ev[W3_NEV]{}; } jx[W3_MAX_JOBS - 1];
    JobState js[W3_MAX_JOBS];
    hipStream_t s_pred = nullptr, s_code[W3_MAX_JOBS] = {};   // created by the first w3_encode_submit
    int next_job = 0, last_job = -1;
    bool pooled_streams = false;
    hipStream_t s_side = nullptr, s_verify[W3_MAX_JOBS] = {};   // the workspaces' side stream (one: predict phases never overlap) and re-prediction streams
    // sharded calls in flight (w3_encode_sharded_submit / w3_encode_sharded_wait): this context's shard of step `slot`, in staging
    // buffers of its own until the step is gathered
    struct ShardSlot { int state = 0, djob = -1; DevBuf out, lens, total; size_t nb = 0, n = 0, cap = 0; const uint8_t *d_in = nullptr; size_t block_size = 0; w3_model_spec spec{}; w3_huff_table huff_copy[W3_MAX_HUFF]; } ss[W3_MAX_JOBS];
    // host-buffer calls in flight (w3_encode_host_submit / w3_encode_host_wait; w3_encode_blocks cuts its input into such calls)
    HostJob hj[W3_MAX_HOST_JOBS];
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;   // where the copies go: the context's stream, or (W3_OPT_TUNE bit 16) the two below
    hipStream_t s_h2d_own = nullptr, s_d2h_own = nullptr;
    uint64_t hseq = 0;
    uint32_t host_chunk_blocks = 0;                 // W3_OPT_HOST_CHUNK_BLOCKS (0 = auto)
    uint32_t n_encodes = 0;                         // rotates the sampled verification over the blocks, whichever job slot a call lands on
};

// the members of job j under one name
struct JobRef {
    TwoPhaseWs &tp; DevBuf &stripes, &flag, &bits, &offs, &huff; hipEvent_t *ev; JobState &st;
};
static JobRef jobref(w3_ctx *c, int j) {
    if (j == 0) return JobRef{c->tp, c->stripes, c->flag, c->bits, c->offs, c->huff, c->ev, c->js[0]};
    w3_ctx::JobWs &x = c->jx[j - 1];
    return JobRef{x.tp, x.stripes, x.flag, x.bits, x.offs, x.huff, x.ev, c->js[j]};
}

#define HIPCHK(ctx, expr)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
            return W3_E_HIP;                                                                    \
        }                                                                                       \
    } while (0)

static int ensure(w3_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return W3_OK;
    if (b.p) { HIPCHK(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = std::max<size_t>(bytes, 256);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        ctx->err = "hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(e);
        (void)hipGetLastError();
        return W3_E_NOMEM;
    }
    b.cap = want;
    return W3_OK;
}
#define ENSURE(ctx, buf, bytes) do { int r_ = ensure(ctx, buf, bytes); if (r_) return r_; } while (0)

// Every entry point but w3_encode_submit / w3_encode_wait (and their host-buffer forms) works on job 0's workspace and the context's
// options: none of them may run while a submitted call is in flight (include/w3hip.h).
static int jobs_idle(w3_ctx *ctx) {
    for (const auto &st : ctx->js)
        if (st.state == 1) { ctx->err = "asynchronous jobs are in flight on this context: w3_encode_wait them first"; return W3_E_INVALID; }
    for (const auto &h : ctx->hj)
        if (h.state != 0) { ctx->err = "host-buffer jobs are in flight on this context: w3_encode_host_wait them first"; return W3_E_INVALID; }
    for (const auto &x : ctx->ss)
        if (x.state != 0) { ctx->err = "sharded jobs are in flight on this context: w3_encode_sharded_wait them first"; return W3_E_INVALID; }
    return W3_OK;
}

extern "C" int w3_abi_version(void) { return W3_ABI_VERSION; }

extern "C" const char *w3_strerror(int code) {
    switch (code) {
    case W3_OK: return "ok";
    case W3_E_INVALID: return "invalid argument or malformed model spec";
    case W3_E_NOSPACE: return "output buffer too small";
    case W3_E_HIP: return "HIP runtime error";
    case W3_E_UNSUPPORTED: return "model spec not implemented on the device";
    case W3_E_NOMEM: return "device workspace does not fit";
    case W3_E_FORMAT: return "bad container magic";
    default: return "unknown error";
    }
}

extern "C" const char *w3_last_error(const w3_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

// The pipeline's streams are kept for the life of the process and handed from context to context (per device and priority): HIP deals
// hardware queues out when a stream is CREATED, by the queues' reference counts at that moment, and a context created after others
// had come and gone got code streams that shared queues (bench.py's later lines ran at the two-in-flight rate with four in flight).
struct StreamPool {
    std::mutex mu;
    std::vector<std::pair<long, hipStream_t>> idle;   // key = device * 8 + class (0 = predict, 1 = code, 2 = side, 3 = verification, 4 = the context's own launch / copy stream)
    hipStream_t take(long key) {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < idle.size(); i++)
            if (idle[i].first == key) { hipStream_t s = idle[i].second; idle.erase(idle.begin() + (long)i); return s; }
        return nullptr;
    }
    void give(long key, hipStream_t s) { std::lock_guard<std::mutex> g(mu); idle.emplace_back(key, s); }
};
static StreamPool &stream_pool() { static StreamPool *p = new StreamPool; return *p; }   // (never destroyed: streams outlive static teardown order)

extern "C" int w3_ctx_create(int device, w3_ctx **out) {
    if (!out) return W3_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return W3_E_HIP;
    if (hipSetDevice(device) != hipSuccess) return W3_E_HIP;
    w3_ctx *c = new w3_ctx();
    c->device = device;
    // A BLOCKING stream: it is ordered against the legacy default (NULL) stream like any ordinary stream, so a caller that
    // produces d_in on the default stream and passes stream = NULL (or torch's default-stream handle, which is 0) gets the
    // order it expects.  The side streams of the predict phase fork from and join the launch stream with events.
    // Taken from the process-wide pool like the pipeline's streams: the host-buffer calls put their PCIe copies on this stream, and a
    // stream created after other contexts have come and gone can land on the hardware queue of a (pooled, still living) predict stream —
    // every copy then holds that call's kernels back (measured: 84.5 ms per 1e9-byte call in flight inside bench.py, where two contexts
    // had lived before, against 71.0 in a fresh process).
    c->stream = stream_pool().take(device * 8L + 4);
    if (!c->stream && hipStreamCreate(&c->stream) != hipSuccess) { delete c; return W3_E_HIP; }
    for (auto &e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete c; return W3_E_HIP; }
    *out = c;
    return W3_OK;
}

extern "C" void w3_ctx_destroy(w3_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    DevBuf *bufs[] = {&ctx->tables, &ctx->stripes, &ctx->lens, &ctx->offs, &ctx->total, &ctx->flag,
                      &ctx->io_in, &ctx->io_out, &ctx->coffs, &ctx->misc, &ctx->cm_luts, &ctx->achash_luts, &ctx->huff, &ctx->bits, &ctx->sweep};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    ctx->tp.release();
    for (auto &x : ctx->jx) {
        x.tp.release();
        DevBuf *bufs1[] = {&x.stripes, &x.flag, &x.bits, &x.offs, &x.huff};
        for (DevBuf *b : bufs1)
            if (b->p) (void)hipFree(b->p);
        for (auto &e : x.ev) if (e) (void)hipEventDestroy(e);
    }
    for (auto &st : ctx->js) {
        if (st.ev_done) (void)hipEventDestroy(st.ev_done);
        if (st.ev_in) (void)hipEventDestroy(st.ev_in);
        if (st.ev_a) (void)hipEventDestroy(st.ev_a);
        if (st.ev_apm) (void)hipEventDestroy(st.ev_apm);
        if (st.h_status) (void)hipHostFree(st.h_status);
    }
    if (ctx->s_side) stream_pool().give(ctx->device * 8L + 2, ctx->s_side);
    for (auto &sv : ctx->s_verify) if (sv) stream_pool().give(ctx->device * 8L + 3, sv);
    if (ctx->pooled_streams) {   // (idle by now: the device was synchronised above)
        if (ctx->s_pred) stream_pool().give(ctx->device * 8L, ctx->s_pred);
        for (auto &sc : ctx->s_code) if (sc) stream_pool().give(ctx->device * 8L + 1, sc);
    } else {
        if (ctx->s_pred) (void)hipStreamDestroy(ctx->s_pred);
        for (auto &sc : ctx->s_code) if (sc) (void)hipStreamDestroy(sc);
    }
    for (auto &x : ctx->ss) {
        DevBuf *bufs3[] = {&x.out, &x.lens, &x.total};
        for (DevBuf *b : bufs3)
            if (b->p) (void)hipFree(b->p);
    }
    for (auto &h : ctx->hj) {
        DevBuf *bufs2[] = {&h.d_in, &h.d_out, &h.d_lens, &h.d_total};
        for (DevBuf *b : bufs2)
            if (b->p) (void)hipFree(b->p);
        if (h.ev_d2h) (void)hipEventDestroy(h.ev_d2h);
    }
    if (ctx->s_h2d_own) (void)hipStreamDestroy(ctx->s_h2d_own);
    if (ctx->s_d2h_own) (void)hipStreamDestroy(ctx->s_d2h_own);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->stream) stream_pool().give(ctx->device * 8L + 4, ctx->stream);   // (idle: the device was synchronised above)
    delete ctx;
}

extern "C" int w3_ctx_set_option(w3_ctx *ctx, int opt, int64_t value) {
    if (!ctx) return W3_E_INVALID;
    // A job in flight has taken its paths, variants and coder from the options: changing them between its predict and code stages would
    // mix two settings in one call.  Two options are exempt: W3_OPT_TIMING (read when a call is submitted) and W3_OPT_TUNE (scheduling
    // only: the output is identical whatever is set).
    if (opt != W3_OPT_TIMING && opt != W3_OPT_TUNE) { const int rc_ = jobs_idle(ctx); if (rc_) return rc_; }
    switch (opt) {
    case W3_OPT_PATH:
        if (value < W3_PATH_AUTO || value > W3_PATH_TWOPHASE) return W3_E_INVALID;
        ctx->opt_path = (int)value;
        return W3_OK;
    case W3_OPT_TIMING: ctx->opt_timing = value ? 1 : 0; return W3_OK;
    case W3_OPT_CODER:
        if (value < 0 || value > 5) return W3_E_INVALID;
        ctx->tp.coder_mode = (int)value;
        return W3_OK;
    case W3_OPT_DEBUG_STAMPS: ctx->tp.debug_stamps = value ? 1 : 0; return W3_OK;
    case W3_OPT_ACC_LIMIT:
        if (value < 19 || value > 46) return W3_E_INVALID;
        ctx->tp.acc_limit = (uint32_t)value;
        return W3_OK;
    case W3_OPT_VARIANT:
        if (value < 0 || value > 2047) return W3_E_INVALID;
        // the fault-injection hook exists for the test of the sampled verification: without the verification it would only corrupt output
        if ((value & W3_VAR_INJECT_LDS_FAULT) && !ctx->tp.verify) { ctx->err = "W3_OPT_VARIANT bit 32 (fault injection) needs W3_OPT_VERIFY on"; return W3_E_INVALID; }
        ctx->tp.variant = (uint32_t)value;
        ctx->tp.lds_order = -1;   // re-run the lane-order self-test under the new setting
        for (auto &x : ctx->jx) x.tp.lds_order = -1;   // (the other job slots follow job 0: sync_job_options)
        return W3_OK;
    case W3_OPT_VERIFY:
        if (!value && (ctx->tp.variant & W3_VAR_INJECT_LDS_FAULT)) { ctx->err = "W3_OPT_VERIFY cannot be switched off while the fault-injection variant is set"; return W3_E_INVALID; }
        if (value < 0 || value > 256) return W3_E_INVALID;
        ctx->tp.verify = (int)value;   // 0 = off, v >= 1: v / 256 of the blocks are re-predicted per call (twophase_verify)
        return W3_OK;
    case W3_OPT_SLOT_BUDGET_MB:
        if (value < 0 || value > (1 << 20)) return W3_E_INVALID;
        ctx->tp.slot_budget_mb = (uint32_t)value;
        return W3_OK;
    case W3_OPT_TUNE:
        if (value < 0 || value > 0xFFFFF) return W3_E_INVALID;
        ctx->tp.tune = (uint32_t)value;
        return W3_OK;
    case W3_OPT_FAULT_BLOCK:
        if (value < -1 || value > 0x7FFFFFFF) return W3_E_INVALID;
        ctx->tp.fault_block = value < 0 ? 0xFFFFFFFFu : (uint32_t)value;
        return W3_OK;
    case W3_OPT_HOST_CHUNK_BLOCKS:
        if (value < 0 || value > 0x7FFFFFFF) return W3_E_INVALID;
        ctx->host_chunk_blocks = (uint32_t)value;
        return W3_OK;
    default: return W3_E_INVALID;
    }
}

extern "C" int w3_get_timing(const w3_ctx *ctx, w3_timing *out) {
    if (!ctx || !out) return W3_E_INVALID;
    *out = ctx->timing;
    return W3_OK;
}

// Hard bound: Counter probabilities lie in [1, 65535] so one bit-step costs at
// most 16 output bits (SURVEY §7 hard part 4): 16 bytes per input byte, plus
// the flush byte(s) per block.
extern "C" size_t w3_max_compressed_size(size_t n, size_t block_size) {
    if (block_size == 0) return 0;
    size_t nb = (n + block_size - 1) / block_size;
    return 16 * n + 8 * nb;
}

// ---------------------------------------------------------------------------
// model spec -> ordered leaf list
// ---------------------------------------------------------------------------
static int parse_spec(const w3_model_spec *spec, ParsedSpec &ps) {
    if (!spec || spec->n_nodes == 0 || spec->n_nodes > W3_MAX_NODES) return W3_E_INVALID;
    if (spec->n_huff > W3_MAX_HUFF || (spec->n_huff && !spec->huff)) return W3_E_INVALID;   // (the spec must be zero-initialised: w3hip.h)
    int depth = 0;
    uint32_t huff_used = 0;   // table sets some W3_HIST_HUFF leaf refers to: only those are read
    ps = ParsedSpec();
    for (uint32_t i = 0; i < spec->n_nodes; i++) {
        const w3_node &nd = spec->nodes[i];
        if (nd.kind == W3_NODE_APM) {
            // APM(model): one input.  Implemented as a chain at the root of the tree only.
            if (depth < 1 || nd.align > W3_APM_ORDER1 || nd.max_bits < 1 || nd.max_bits > 15) return W3_E_INVALID;
            if (depth != 1) return W3_E_UNSUPPORTED;
            if (ps.n_apm == W3_MAX_APM) return W3_E_UNSUPPORTED;
            ps.apm[ps.n_apm++] = nd;
            continue;
        }
        if (ps.n_apm) return nd.kind == W3_NODE_ORDERN || nd.kind == W3_NODE_SLOT_STATE || nd.kind == W3_NODE_BEST_OF_TWO ? W3_E_UNSUPPORTED : W3_E_INVALID;
        if (nd.kind == W3_NODE_ORDERN) {
            // OrderN::new allocates 1<<bits counters; masks are u32/u8 (ordern.rs:35-43)
            if (nd.bits < 1 || nd.bits > 32 || nd.align > 7 || nd.align > nd.bits) return W3_E_INVALID;
            if ((int)nd.bits - (int)nd.align > 31) return W3_E_INVALID;
            if (nd.history > W3_HIST_HUFF) return W3_E_INVALID;
            if (nd.history == W3_HIST_AC && nd.max_bits > 32) return W3_E_INVALID;
            if (nd.history == W3_HIST_HUFF) {
                if (nd.reserved >= spec->n_huff) return W3_E_INVALID;
                huff_used |= 1u << nd.reserved;
            }
            if (ps.n_leaves == W3_MAX_LEAVES) return W3_E_UNSUPPORTED;
            ps.leaf[ps.n_leaves++] = nd;
            depth++;
        } else if (nd.kind == W3_NODE_SLOT_STATE) {
            if (nd.bits > 7 || nd.log_cells < 1 || nd.log_cells > 24 || nd.frozen) return W3_E_INVALID;
            if (ps.n_leaves == W3_MAX_LEAVES) return W3_E_UNSUPPORTED;
            ps.leaf[ps.n_leaves++] = nd;
            ps.has_slot = true;
            depth++;
        } else if (nd.kind == W3_NODE_BEST_OF_TWO) {
            if (depth < 2) return W3_E_INVALID;
            depth--;
        } else {
            return W3_E_INVALID;
        }
    }
    // the table sets travel to the device only when a leaf uses one (a spec without HuffHistory leaves never has its huff
    // pointer dereferenced); code lengths index shifts of u32 values
    ps.n_huff = huff_used ? spec->n_huff : 0;
    ps.huff = ps.n_huff ? spec->huff : nullptr;
    for (uint32_t k = 0; k < ps.n_huff; k++) {
        if (!((huff_used >> k) & 1u)) continue;
        for (int v = 0; v < 256; v++)
            if (ps.huff[k].len[v] > 16 || ps.huff[k].rem_len[v] > 16) return W3_E_INVALID;
    }
    return depth == 1 ? W3_OK : W3_E_INVALID;
}

// HuffHistory table sets of the spec -> device (per call: the tables are the caller's memory); job 0's copy
static int stage_huff(w3_ctx *ctx, hipStream_t s, const ParsedSpec &ps);

extern "C" int w3_spec_validate(const w3_model_spec *spec) {
    ParsedSpec ps;
    return parse_spec(spec, ps);
}

static uint64_t next_pow2(uint64_t v) {
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Lay out the per-lane model tables of the generic path.
static uint64_t layout_generic(const ParsedSpec &ps, size_t block_size, GenericArgs &ga) {
    const uint64_t steps = (uint64_t)block_size * 8;
    const uint64_t hash_slots = std::max<uint64_t>(1024, next_pow2(2 * steps));
    const uint64_t hash_bytes = hash_slots * 8;
    uint64_t off = 0;
    ga.n_leaves = ps.n_leaves;
    for (int l = 0; l < ps.n_leaves; l++) {
        const w3_node &nd = ps.leaf[l];
        LeafParam &lp = ga.leaf[l];
        memset(&lp, 0, sizeof lp);
        lp.bits = nd.bits; lp.align = nd.align; lp.hist = nd.history; lp.max_bits = nd.max_bits; lp.frozen = nd.frozen;
        lp.huff_idx = nd.history == W3_HIST_HUFF ? nd.reserved : 0;
        memcpy(lp.table, nd.table, sizeof lp.table);
        lp.tbl_off = off;
        if (nd.kind == W3_NODE_SLOT_STATE) {   // HashMap of 2^log_cells 96-byte Cells (hashmap.rs:7-22)
            lp.kind = 1; lp.order = nd.bits; lp.log_cells = nd.log_cells;
            off += 96ull << nd.log_cells;
            continue;
        }
        lp.hist_mask = (uint32_t)((1ull << (nd.bits - nd.align)) - 1ull);
        if (nd.frozen) continue;
        const uint64_t direct_bytes = 4ull << nd.bits;
        if (direct_bytes <= hash_bytes) { lp.use_hash = 0; off += direct_bytes; }
        else { lp.use_hash = 1; lp.hash_mask = (uint32_t)(hash_slots - 1); off += hash_bytes; }
    }
    return std::max<uint64_t>(off, 16);
}

// ACHistory leaves of the lane-per-block kernels: tabulate the coder states of every 16-bit history prefix once per call
// (k_achash_lut, w3_predict.h) so that leaf_ctx looks the hash up instead of running the nested coder bit by bit.
static int prepare_achash_luts(w3_ctx *ctx, hipStream_t s, GenericArgs &ga);

// ---------------------------------------------------------------------------
// pack: scan block lengths, compact stripes into d_out
// ---------------------------------------------------------------------------
static int run_pack(w3_ctx *ctx, JobRef &J, hipStream_t s, const uint8_t *stripes, uint64_t stride, const uint32_t *d_lens, uint32_t nb,
                    uint8_t *d_out, size_t out_cap, uint64_t *d_total) {
    ENSURE(ctx, J.offs, (size_t)nb * 8);
    hipLaunchKernelGGL(k_scan_lens, dim3(1), dim3(1024), 0, s, d_lens, (uint64_t *)J.offs.p, d_total, nb);
    uint32_t grid = std::min<uint32_t>(nb, 256 * 8);
    hipLaunchKernelGGL(k_pack, dim3(grid), dim3(256), 0, s, stripes, stride, d_lens, (const uint64_t *)J.offs.p, d_out,
                       (uint64_t)out_cap, nb);
    HIPCHK(ctx, hipGetLastError());
    return W3_OK;
}

// ---------------------------------------------------------------------------
// generic path
// ---------------------------------------------------------------------------
static int table_budget(w3_ctx *ctx, uint64_t lane_stride, uint32_t want_lanes, uint32_t &lanes_out) {
    size_t free_b = 0, total_b = 0;
    HIPCHK(ctx, hipMemGetInfo(&free_b, &total_b));
    // (Every block of a call resident at once matters to the decoders: they are latency chains per block, so two batches take twice as
    // long as one.  The default model's tables are 10 MiB per block = 153 GB at enwik9 size.)
    const uint64_t avail_b = (uint64_t)free_b + ctx->tables.cap;
    uint64_t budget = std::min<uint64_t>(avail_b > (12ull << 30) ? avail_b - (12ull << 30) : avail_b / 2, 224ull << 30);
    uint64_t lanes = budget / lane_stride;
    if (lanes >= want_lanes) lanes = want_lanes;
    else lanes = lanes / 64 * 64;
    if (lanes == 0) {
        ctx->err = "model tables of one wavefront (" + std::to_string(lane_stride * 64) + " B) exceed the device budget";
        return W3_E_NOMEM;
    }
    // One hipMalloc of that size can still fail (fragmentation, another process or rank on the same GPU, a caching allocator that
    // holds what hipMemGetInfo calls free): take half the lanes then — the callers run batches — rather than fail the call.
    for (;;) {
        const int rc = ensure(ctx, ctx->tables, (size_t)lanes * lane_stride);
        if (rc == W3_OK) break;
        if (rc != W3_E_NOMEM || lanes <= 64) return rc;
        lanes = std::max<uint64_t>(64, lanes / 2 / 64 * 64);
    }
    lanes_out = (uint32_t)lanes;
    return W3_OK;
}

static int prepare_achash_luts(w3_ctx *ctx, hipStream_t s, GenericArgs &ga) {
    int n_ac = 0;
    for (int l = 0; l < ga.n_leaves; l++) n_ac += ga.leaf[l].kind == 0 && ga.leaf[l].hist == W3_HIST_AC && !ga.leaf[l].frozen;
    if (!n_ac) return W3_OK;
    const size_t entries = (size_t)8u << W3_ACHASH_LUT_BITS, per_leaf = entries * 18;
    ENSURE(ctx, ctx->achash_luts, per_leaf * (size_t)n_ac);
    int k = 0;
    for (int l = 0; l < ga.n_leaves; l++) {
        LeafParam &lp = ga.leaf[l];
        if (!(lp.kind == 0 && lp.hist == W3_HIST_AC && !lp.frozen)) continue;
        uint8_t *base = (uint8_t *)ctx->achash_luts.p + per_leaf * (size_t)k++;
        HashArgs ha;
        memset(&ha, 0, sizeof ha);
        ha.max_bits = lp.max_bits; ha.hmask = 0xFFu;
        memcpy(ha.table, lp.table, sizeof ha.table);
        ha.lut = (uint4 *)base; ha.lut_key = (uint16_t *)(base + entries * 16);
        hipLaunchKernelGGL(k_achash_lut, dim3((unsigned)(entries / 256)), dim3(256), 0, s, ha);
        lp.lut = (const uint4 *)base;
    }
    HIPCHK(ctx, hipGetLastError());
    return W3_OK;
}

// k_decode_spec's group: the whole nibble.  Half a nibble (W3_OPT_TUNE bit 14: four lanes per block, 6 instead of 15 speculative look-ups per
// nibble and leaf) was measured for the large batches, which are bound by those look-ups' HBM traffic — order012apm 736 -> 758 MiB/s at 1e9 B,
// but order012 856 -> 778, Order0 3,906 -> 3,366, main.rs default 1,585 -> 1,417, and 189 -> 104 MiB/s at 1e8 B: four round trips per nibble
// instead of two, and a quarter of the wavefronts to hide them (profiles/r3_decode_spec/).  Kept as a tested variant.
static int decode_group_bits(const w3_ctx *ctx, uint32_t blocks_in_batch) {
    (void)blocks_in_batch;
    return (ctx->tp.tune & 16384u) ? 2 : 4;
}

static int generic_encode(w3_ctx *ctx, hipStream_t s, const ParsedSpec &ps, const uint8_t *d_in, size_t n, size_t block_size,
                          uint32_t nb, uint32_t stripe_cap, uint32_t *d_lens) {
    GenericArgs ga;
    memset(&ga, 0, sizeof ga);
    const uint64_t lane_stride = layout_generic(ps, block_size, ga);
    uint32_t lanes = 0;
    int rc = table_budget(ctx, lane_stride, nb, lanes);
    if (rc) return rc;
    if ((rc = prepare_achash_luts(ctx, s, ga))) return rc;
    ENSURE(ctx, ctx->tables, (size_t)lanes * lane_stride);
    ga.n = n; ga.block_size = (uint32_t)block_size;
    ga.huff = ctx->tp.huff; ga.n_huff = (int)ps.n_huff;
    ga.tables = (uint8_t *)ctx->tables.p; ga.lane_stride = lane_stride;
    ga.in = d_in; ga.stripe_cap = stripe_cap; ga.out_len = d_lens; ga.overflow = (uint32_t *)ctx->flag.p;
    ga.out_bits = (uint32_t *)ctx->bits.p;
    for (uint32_t first = 0; first < nb; first += lanes) {
        uint32_t cnt = std::min(lanes, nb - first);
        ga.first_block = first; ga.n_lanes = cnt;
        ga.stripes = (uint8_t *)ctx->stripes.p + (uint64_t)first * stripe_cap;
        HIPCHK(ctx, hipMemsetAsync(ctx->tables.p, 0, (size_t)cnt * lane_stride, s));
        switch (ga.n_leaves) {   // 1-4 leaves: all Counter loads of a step in flight together
        case 1: hipLaunchKernelGGL((k_generic_nl<false, 1>), dim3((cnt + 63) / 64), dim3(64), 0, s, ga); break;
        case 2: hipLaunchKernelGGL((k_generic_nl<false, 2>), dim3((cnt + 63) / 64), dim3(64), 0, s, ga); break;
        case 3: hipLaunchKernelGGL((k_generic_nl<false, 3>), dim3((cnt + 63) / 64), dim3(64), 0, s, ga); break;
        case 4: hipLaunchKernelGGL((k_generic_nl<false, 4>), dim3((cnt + 63) / 64), dim3(64), 0, s, ga); break;
        default: hipLaunchKernelGGL(k_generic<false>, dim3((cnt + 63) / 64), dim3(64), 0, s, ga); break;
        }
        HIPCHK(ctx, hipGetLastError());
    }
    return W3_OK;
}

static int generic_decode(w3_ctx *ctx, hipStream_t s, const ParsedSpec &ps, const uint8_t *d_cin, const uint32_t *d_lens, uint32_t nb,
                          size_t block_size, uint64_t orig_len, uint8_t *d_out) {
    GenericArgs ga;
    memset(&ga, 0, sizeof ga);
    const uint64_t lane_stride = layout_generic(ps, block_size, ga);
    uint32_t lanes = 0;
    int rc = table_budget(ctx, lane_stride, nb, lanes);
    if (rc) return rc;
    if ((rc = prepare_achash_luts(ctx, s, ga))) return rc;
    ENSURE(ctx, ctx->tables, (size_t)lanes * lane_stride);
    ENSURE(ctx, ctx->coffs, (size_t)nb * 8);
    ENSURE(ctx, ctx->total, 8);
    hipLaunchKernelGGL(k_scan_lens, dim3(1), dim3(1024), 0, s, d_lens, (uint64_t *)ctx->coffs.p, (uint64_t *)ctx->total.p, nb);
    ga.n = orig_len; ga.block_size = (uint32_t)block_size;
    ga.huff = ctx->tp.huff; ga.n_huff = (int)ps.n_huff;
    ga.tables = (uint8_t *)ctx->tables.p; ga.lane_stride = lane_stride;
    ga.cin = d_cin; ga.coffs = (const uint64_t *)ctx->coffs.p; ga.clens = d_lens; ga.dout = d_out;
    for (uint32_t first = 0; first < nb; first += lanes) {
        uint32_t cnt = std::min(lanes, nb - first);
        ga.first_block = first; ga.n_lanes = cnt;
        HIPCHK(ctx, hipMemsetAsync(ctx->tables.p, 0, (size_t)cnt * lane_stride, s));
        {   // the nibble's context tree at once, sixteen lanes per block (w3_decode_spec.h), where it applies
            CmArgs ca;
            memset(&ca, 0, sizeof ca);
            ca.g = ga;
            ca.dflags = (ctx->tp.tune >> 17) & 7u;
            if (decode_spec_covers(ca) && !(ctx->tp.variant & W3_VAR_DECODE_LANE)) {
                launch_decode_spec(ca, cnt, s, decode_group_bits(ctx, cnt));
                HIPCHK(ctx, hipGetLastError());
                continue;
            }
        }
        switch (ga.n_leaves) {
        case 1: hipLaunchKernelGGL((k_generic_nl<true, 1>), dim3((cnt + 63) / 64), dim3(64), 0, s, ga); break;
        case 2: hipLaunchKernelGGL((k_generic_nl<true, 2>), dim3((cnt + 63) / 64), dim3(64), 0, s, ga); break;
        case 3: hipLaunchKernelGGL((k_generic_nl<true, 3>), dim3((cnt + 63) / 64), dim3(64), 0, s, ga); break;
        case 4: hipLaunchKernelGGL((k_generic_nl<true, 4>), dim3((cnt + 63) / 64), dim3(64), 0, s, ga); break;
        default: hipLaunchKernelGGL(k_generic<true>, dim3((cnt + 63) / 64), dim3(64), 0, s, ga); break;
        }
        HIPCHK(ctx, hipGetLastError());
    }
    return W3_OK;
}

// ---------------------------------------------------------------------------
// CM path (slot-state leaves and/or APM chain): k_cm, lane per block
// ---------------------------------------------------------------------------
static int cm_luts(w3_ctx *ctx, hipStream_t s, CmArgs &ca) {
    const size_t st_bytes = (size_t)kStSize * 8, str_bytes = 4096 * 2, sq_bytes = 4096 * 2;
    if (!ctx->cm_luts.p) {
        ENSURE(ctx, ctx->cm_luts, st_bytes + str_bytes + sq_bytes);
        std::vector<StEntry> t(kStSize);
        build_state_table(t.data());
        std::vector<uint32_t> packed(2 * kStSize);
        for (int i = 0; i < kStSize; i++) {
            packed[2 * i] = t[i].prob | ((uint32_t)t[i].next0 << 16);
            packed[2 * i + 1] = t[i].next1 | ((uint32_t)t[i].conf << 16);
        }
        std::vector<int16_t> str(4096);
        std::vector<uint16_t> sq(4096);
        build_stretch_squash(str.data(), sq.data());
        uint8_t *d = (uint8_t *)ctx->cm_luts.p;
        HIPCHK(ctx, hipMemcpy(d, packed.data(), st_bytes, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(d + st_bytes, str.data(), str_bytes, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(d + st_bytes + str_bytes, sq.data(), sq_bytes, hipMemcpyHostToDevice));
    }
    (void)s;
    uint8_t *d = (uint8_t *)ctx->cm_luts.p;
    ca.st = (const uint2 *)d;
    ca.stretch = (const int16_t *)(d + st_bytes);
    ca.squash = (const uint16_t *)(d + st_bytes + str_bytes);
    return W3_OK;
}

static uint64_t layout_cm(const ParsedSpec &ps, size_t block_size, CmArgs &ca) {
    uint64_t off = (layout_generic(ps, block_size, ca.g) + 15) / 16 * 16;
    ca.n_apm = ps.n_apm;
    for (int k = 0; k < ps.n_apm; k++) {
        ca.apm[k].ctx_kind = ps.apm[k].align;
        ca.apm[k].rate = ps.apm[k].max_bits;
        ca.apm[k].off = off;
        // (272 rows per page of 256: k_decode_spec keeps the table nibble-major, 17 groups of 16 node columns — w3_decode_spec.h)
        off += ((ps.apm[k].align == W3_APM_ORDER1 ? 256ull : 1ull) * 272 * 33 * 2 + 15) / 16 * 16;
    }
    return off;
}

template <bool DECODE>
static int cm_run(w3_ctx *ctx, hipStream_t s, CmArgs &ca, uint64_t lane_stride, uint32_t nb, uint32_t stripe_cap) {
    uint32_t lanes = 0;
    int rc = table_budget(ctx, lane_stride, nb, lanes);
    if (rc) return rc;
    ENSURE(ctx, ctx->tables, (size_t)lanes * lane_stride);
    if ((rc = cm_luts(ctx, s, ca))) return rc;
    ca.g.tables = (uint8_t *)ctx->tables.p; ca.g.lane_stride = lane_stride;
    for (uint32_t first = 0; first < nb; first += lanes) {
        uint32_t cnt = std::min(lanes, nb - first);
        ca.g.first_block = first; ca.g.n_lanes = cnt;
        if (!DECODE) ca.g.stripes = (uint8_t *)ctx->stripes.p + (uint64_t)first * stripe_cap;
        HIPCHK(ctx, hipMemsetAsync(ctx->tables.p, 0, (size_t)cnt * lane_stride, s));
        const bool spec_dec = DECODE && decode_spec_covers(ca) && !(ctx->tp.variant & W3_VAR_DECODE_LANE);
        ca.dflags = (ctx->tp.tune >> 17) & 7u;
        const bool nm_tables = spec_dec && decode_spec_nibble_major(ca, cnt, decode_group_bits(ctx, cnt));   // (the APM tables' layout follows the kernel's)
        for (int k = 0; k < ca.n_apm; k++)
            hipLaunchKernelGGL(k_cm_init_apm, dim3(2048), dim3(256), 0, s, ca.g.tables, lane_stride, ca.apm[k].off,
                               nm_tables ? (ca.apm[k].ctx_kind ? 256u * 272u : 272u) : (ca.apm[k].ctx_kind ? 65536u : 256u), cnt, ca.squash, nm_tables ? 1u : 0u);
        bool has_slot = false;
        for (int l = 0; l < ca.g.n_leaves; l++) has_slot |= ca.g.leaf[l].kind == 1;
        const dim3 grid((cnt + 63) / 64), blk(64);
        if (spec_dec) launch_decode_spec(ca, cnt, s, decode_group_bits(ctx, cnt));   // (w3_decode_spec.h)
        else if (!has_slot && ca.g.n_leaves <= 4) {   // Counter leaves + APM chain: all Counter loads of a step in flight together
            switch (ca.g.n_leaves) {
            case 1: hipLaunchKernelGGL((k_cm_nl<DECODE, 1>), grid, blk, 0, s, ca); break;
            case 2: hipLaunchKernelGGL((k_cm_nl<DECODE, 2>), grid, blk, 0, s, ca); break;
            case 3: hipLaunchKernelGGL((k_cm_nl<DECODE, 3>), grid, blk, 0, s, ca); break;
            default: hipLaunchKernelGGL((k_cm_nl<DECODE, 4>), grid, blk, 0, s, ca); break;
            }
        } else {
            int n_slot = 0;
            for (int l = 0; l < ca.g.n_leaves; l++) n_slot += ca.g.leaf[l].kind == 1;
            if (n_slot <= W3_CM_STAGED_MAX && !(ctx->tp.variant & W3_VAR_CM_UNSTAGED)) hipLaunchKernelGGL(k_cm_staged<DECODE>, grid, blk, 0, s, ca);   // slot cells staged in LDS
            else hipLaunchKernelGGL(k_cm<DECODE>, grid, blk, 0, s, ca);
        }
        HIPCHK(ctx, hipGetLastError());
    }
    return W3_OK;
}

static int cm_encode(w3_ctx *ctx, hipStream_t s, const ParsedSpec &ps, const uint8_t *d_in, size_t n, size_t block_size,
                     uint32_t nb, uint32_t stripe_cap, uint32_t *d_lens) {
    CmArgs ca;
    memset(&ca, 0, sizeof ca);
    const uint64_t lane_stride = layout_cm(ps, block_size, ca);
    { int rc_ = prepare_achash_luts(ctx, s, ca.g); if (rc_) return rc_; }
    ca.g.n = n; ca.g.block_size = (uint32_t)block_size;
    ca.g.huff = ctx->tp.huff; ca.g.n_huff = (int)ps.n_huff;
    ca.g.in = d_in; ca.g.stripe_cap = stripe_cap; ca.g.out_len = d_lens; ca.g.overflow = (uint32_t *)ctx->flag.p;
    ca.g.out_bits = (uint32_t *)ctx->bits.p;
    return cm_run<false>(ctx, s, ca, lane_stride, nb, stripe_cap);
}

static int cm_decode(w3_ctx *ctx, hipStream_t s, const ParsedSpec &ps, const uint8_t *d_cin, const uint32_t *d_lens, uint32_t nb,
                     size_t block_size, uint64_t orig_len, uint8_t *d_out) {
    CmArgs ca;
    memset(&ca, 0, sizeof ca);
    const uint64_t lane_stride = layout_cm(ps, block_size, ca);
    { int rc_ = prepare_achash_luts(ctx, s, ca.g); if (rc_) return rc_; }
    ENSURE(ctx, ctx->coffs, (size_t)nb * 8);
    ENSURE(ctx, ctx->total, 8);
    hipLaunchKernelGGL(k_scan_lens, dim3(1), dim3(1024), 0, s, d_lens, (uint64_t *)ctx->coffs.p, (uint64_t *)ctx->total.p, nb);
    ca.g.n = orig_len; ca.g.block_size = (uint32_t)block_size;
    ca.g.huff = ctx->tp.huff; ca.g.n_huff = (int)ps.n_huff;
    ca.g.cin = d_cin; ca.g.coffs = (const uint64_t *)ctx->coffs.p; ca.g.clens = d_lens; ca.g.dout = d_out;
    return cm_run<true>(ctx, s, ca, lane_stride, nb, 0);
}

// ---------------------------------------------------------------------------
// encode (device-resident)
// ---------------------------------------------------------------------------
static uint32_t default_stripe_cap(size_t block_size) {
    // realistic bound 2N+64 (adaptive Counter regret is small); the exact bound 16N+8 is the retry size
    uint64_t c = 2 * (uint64_t)block_size + 64;
    return (uint32_t)((c + 15) / 16 * 16);
}
static uint32_t worst_stripe_cap(size_t block_size) {
    uint64_t c = 16 * (uint64_t)block_size + 16;
    return (uint32_t)((c + 15) / 16 * 16);
}

static int check_args(w3_ctx *ctx, size_t n, size_t block_size, bool one_device = true) {
    if (!ctx) return W3_E_INVALID;
    if (block_size == 0 || block_size > (1u << 28)) { ctx->err = "block_size must be in 1..2^28"; return W3_E_INVALID; }
    if ((n + block_size - 1) / block_size > 0x7FFFFFFFull) { ctx->err = "too many blocks"; return W3_E_INVALID; }
    // One call handles less than 4 GiB: the per-byte kernels are launched with one work-item per input byte, and a dispatch counts its
    // work-items in 32 bits (a larger launch would silently cover n mod 2^32 bytes).  Blocks are independent: larger inputs are split by the caller.
    if (one_device && n >= (1ull << 32) - 4096u) { ctx->err = "one call handles less than 4 GiB of input: split larger inputs at block boundaries (blocks are independent)"; return W3_E_UNSUPPORTED; }
    return W3_OK;
}

static float elapsed_ev(hipEvent_t *ev, int slot) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev[2 * slot], ev[2 * slot + 1]) != hipSuccess) { (void)hipGetLastError(); return 0.f; }
    return ms;
}

// per-kernel launch times of one two-phase encode from its events (W3_OPT_TIMING)
static void collect_timing(hipEvent_t *ev, const TwoPhaseWs &ws, bool has_apm, bool has_slot, bool packed, w3_timing &t) {
    t.predict_ms = elapsed_ev(ev, W3_EV_PREDICT); t.coder_ms = elapsed_ev(ev, W3_EV_CODER);
    if (has_apm) t.apm_ms = elapsed_ev(ev, W3_EV_APM);
    if (has_slot) t.slot_ms = elapsed_ev(ev, W3_EV_SLOT);
    if (ws.achash_timed) t.achash_ms = elapsed_ev(ev, W3_EV_ACHASH);
    t.n_wide = (uint32_t)std::min(ws.n_wide, 4);
    for (int w = 0; w < ws.n_wide && w < 4; w++) { t.part_ms[w] = elapsed_ev(ev, W3_EV_PART0 + w); t.rank_ms[w] = elapsed_ev(ev, W3_EV_RANK0 + w); }
    if (ws.small_timed) t.small_ms = elapsed_ev(ev, W3_EV_SMALL);
    t.pack_ms = packed ? elapsed_ev(ev, W3_EV_PACK) : 0.f;
    t.total_ms = elapsed_ev(ev, W3_EV_TOTAL);
}

static int stage_huff_job(w3_ctx *ctx, JobRef &J, hipStream_t s, const ParsedSpec &ps) {
    J.tp.huff = nullptr;
    if (!ps.n_huff) return W3_OK;
    ENSURE(ctx, J.huff, sizeof(w3_huff_table) * W3_MAX_HUFF);
    HIPCHK(ctx, hipMemcpyAsync(J.huff.p, ps.huff, sizeof(w3_huff_table) * ps.n_huff, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipStreamSynchronize(s));   // (pageable source: the caller may free it after the call)
    J.tp.huff = (const w3_huff_table *)J.huff.p;
    return W3_OK;
}

static int stage_huff(w3_ctx *ctx, hipStream_t s, const ParsedSpec &ps) {
    JobRef J = jobref(ctx, 0);
    return stage_huff_job(ctx, J, s, ps);
}

// options live in job 0's workspace (w3_ctx_set_option); job 1 follows it
static void sync_job_options(w3_ctx *ctx, JobRef &J) {
    if (&J.tp == &ctx->tp) return;
    J.tp.coder_mode = ctx->tp.coder_mode; J.tp.acc_limit = ctx->tp.acc_limit; J.tp.debug_stamps = 0;
    J.tp.variant = ctx->tp.variant; J.tp.slot_budget_mb = ctx->tp.slot_budget_mb; J.tp.verify = ctx->tp.verify;
    J.tp.stretch = ctx->tp.stretch; J.tp.squash = ctx->tp.squash; J.tp.st = ctx->tp.st; J.tp.fault_block = ctx->tp.fault_block; J.tp.tune = ctx->tp.tune;
    // the lane-order self-test runs once, on job 0's workspace; the other slots take its verdict (and lose a stale one when
    // W3_OPT_VARIANT has reset job 0's)
    if (ctx->tp.lds_order < 0) (void)twophase_lds_order_ok(ctx->tp, ctx->stream);
    J.tp.lds_order = ctx->tp.lds_order;
}

// The side stream of the predict phase and the job's re-prediction stream: the context's (taken from the process-wide pool), so that
// a context does not create streams — and with them hardware-queue assignments — of its own for every workspace.
static int attach_aux_streams(w3_ctx *ctx, JobRef &J, int j) {
    if (!ctx->s_side) {
        ctx->s_side = stream_pool().take(ctx->device * 8L + 2);
        if (!ctx->s_side) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_side, hipStreamNonBlocking));
    }
    if (!ctx->s_verify[j]) {
        ctx->s_verify[j] = stream_pool().take(ctx->device * 8L + 3);
        if (!ctx->s_verify[j]) {
            int lo_p = 0, hi_p = 0;
            HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
            HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->s_verify[j], hipStreamNonBlocking, lo_p));
        }
    }
    J.tp.side = ctx->s_side; J.tp.vstream = ctx->s_verify[j]; J.tp.ext_streams = true;
    return W3_OK;
}

// d_out == nullptr: counting-sink mode (ACStats, helpers.rs:60-90) — the streams are coded into the stripes as usual, the
// pack is skipped and only J.bits (per-block bit counts) is of interest; d_block_lens may then be a scratch buffer.
// Synchronous: returns when the output is complete.  J = the job whose workspace is used (job 0 for every synchronous entry point;
// w3_encode_wait redoes a job of its own here).
static int encode_core(w3_ctx *ctx, JobRef J, const w3_model_spec *spec, const uint8_t *d_in, size_t n, size_t block_size,
                       uint8_t *d_out, size_t out_cap, uint32_t *d_block_lens, uint64_t *d_total, void *stream) {
    int rc = check_args(ctx, n, block_size);
    if (rc) return rc;
    ParsedSpec ps;
    if ((rc = parse_spec(spec, ps))) { ctx->err = "malformed model spec"; return rc; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const uint32_t nb = (uint32_t)((n + block_size - 1) / block_size);
    memset(&ctx->timing, 0, sizeof ctx->timing);
    ENSURE(ctx, ctx->total, 8);
    uint64_t *total_p = d_total ? d_total : (uint64_t *)ctx->total.p;
    if (nb == 0) {
        HIPCHK(ctx, hipMemsetAsync(total_p, 0, 8, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        return W3_OK;
    }
    if (!d_in || !d_block_lens) return W3_E_INVALID;
    ENSURE(ctx, J.flag, 16);
    ENSURE(ctx, J.bits, (size_t)nb * 4);
    if ((rc = stage_huff_job(ctx, J, s, ps))) return rc;

    bool two = twophase_supported(ps, block_size, n);   // Counter and slot-state leaves + APM chain (decode: k_generic / k_cm)
    if (ctx->opt_path == W3_PATH_GENERIC) two = false;
    if (two && ps.is_cm()) {
        CmArgs lut;
        if ((rc = cm_luts(ctx, s, lut))) return rc;
        ctx->tp.stretch = lut.stretch; ctx->tp.squash = lut.squash; ctx->tp.st = lut.st;
    }
    if (ctx->opt_path == W3_PATH_TWOPHASE && !two) { ctx->err = "spec/block size not covered by the two-phase path"; return W3_E_UNSUPPORTED; }
    if (!two && &J.tp != &ctx->tp) { ctx->err = "internal: the lane-per-block path runs on job 0"; return W3_E_INVALID; }
    sync_job_options(ctx, J);
    if ((rc = attach_aux_streams(ctx, J, &J.tp == &ctx->tp ? 0 : (int)(&J.st - ctx->js)))) return rc;
    J.tp.half_cu = (ctx->tp.variant & W3_VAR_HALF_CU) != 0;
    J.tp.verify_calls = ctx->n_encodes++;

    hipEvent_t *evp = ctx->opt_timing ? J.ev : nullptr;
    uint32_t cap = default_stripe_cap(block_size);
    w3_timing ptm;
    memset(&ptm, 0, sizeof ptm);
    bool cap_raised = false, fault_seen = false;
    uint32_t lds_faults = 0;
    for (int attempt = 0; attempt < 4; attempt++) {
        ENSURE(ctx, J.stripes, (size_t)nb * cap);
        HIPCHK(ctx, hipMemsetAsync(J.flag.p, 0, 16, s));
        if (evp) HIPCHK(ctx, hipEventRecord(evp[2 * W3_EV_TOTAL], s));
        uint32_t fl[4] = {0, 0, 0, 0};
        if (two) {
            J.tp.out_bits = (uint32_t *)J.bits.p;
            J.tp.order_fault = (uint32_t *)J.flag.p + 2;
#ifdef W3_TUNING
            J.tp.apm_oob = (uint32_t *)J.flag.p + 3;
#endif
            memset(&ptm, 0, sizeof ptm);
            rc = twophase_encode(J.tp, s, s, ps, d_in, n, block_size, nb, (uint8_t *)J.stripes.p, cap, d_block_lens, (uint32_t *)J.flag.p, evp, &ptm, ctx->err);
            if (rc) return rc;
            ctx->timing.path = W3_PATH_TWOPHASE;
        } else {
            if (evp) HIPCHK(ctx, hipEventRecord(evp[2 * W3_EV_PREDICT], s));
            rc = ps.is_cm() ? cm_encode(ctx, s, ps, d_in, n, block_size, nb, cap, d_block_lens)
                            : generic_encode(ctx, s, ps, d_in, n, block_size, nb, cap, d_block_lens);
            if (evp) HIPCHK(ctx, hipEventRecord(evp[2 * W3_EV_PREDICT + 1], s));
            ctx->timing.path = W3_PATH_GENERIC;
            if (rc) return rc;
        }
        HIPCHK(ctx, hipMemcpyAsync(fl, J.flag.p, sizeof fl, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        if (two && fl[2]) {
            // sampled verification of the LDS-add rounds (twophase_verify): a mismatch means the hardware did not resolve returning LDS
            // adds in lane order under this load.  The streams just coded cannot be trusted: code the call again with the ballot
            // rounds, and keep this context on them.
            if (fault_seen) { ctx->err = "predict streams differ from their ballot-round re-prediction even without LDS-add rounds (internal error)"; return W3_E_HIP; }
            fault_seen = true;
            lds_faults += fl[2];
            ctx->tp.variant |= W3_VAR_NO_LDS_ATOMICS; ctx->tp.lds_order = 0;
            for (auto &x : ctx->jx) { x.tp.variant |= W3_VAR_NO_LDS_ATOMICS; x.tp.lds_order = 0; }
            ctx->timing.n_recoded_blocks = 0;
            continue;
        }
        if (two && fl[1]) {   // blocks the fast coder handed back (pending-bit run longer than its accumulator)
            ctx->timing.n_recoded_blocks += fl[1];
            rc = twophase_recode(J.tp, s, d_in, n, block_size, nb, (uint8_t *)J.stripes.p, cap, d_block_lens, (uint32_t *)J.flag.p, fl[1], ctx->err);
            if (rc) return rc;
            HIPCHK(ctx, hipMemcpyAsync(fl, J.flag.p, sizeof fl, hipMemcpyDeviceToHost, s));
            HIPCHK(ctx, hipStreamSynchronize(s));
        }
        if (fl[0] & 2u) { ctx->err = "coder pipeline timeout (internal error)"; return W3_E_HIP; }
        if (fl[3]) { ctx->err = "APM kernel: " + std::to_string(fl[3]) + " stores outside the stage's stream and the sink (W3_TUNING store guard)"; return W3_E_HIP; }
        if (!(fl[0] & 1u)) break;
        if (cap_raised) { ctx->err = "stripe overflow at the worst-case bound (internal error)"; return W3_E_HIP; }
        cap = worst_stripe_cap(block_size);  // rare: a block expanded past 2N+64
        cap_raised = true;
        ctx->timing.n_recoded_blocks = 0;
    }
    ctx->timing.n_lds_faults = lds_faults;
    uint64_t total = 0;
    if (d_out) {
        if (evp) HIPCHK(ctx, hipEventRecord(evp[2 * W3_EV_PACK], s));
        rc = run_pack(ctx, J, s, (const uint8_t *)J.stripes.p, cap, d_block_lens, nb, d_out, out_cap, total_p);
        if (evp) { HIPCHK(ctx, hipEventRecord(evp[2 * W3_EV_PACK + 1], s)); HIPCHK(ctx, hipEventRecord(evp[2 * W3_EV_TOTAL + 1], s)); }
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(&total, total_p, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
    } else {
        if (evp) HIPCHK(ctx, hipEventRecord(evp[2 * W3_EV_TOTAL + 1], s));
        HIPCHK(ctx, hipStreamSynchronize(s));
    }
    if (two) {
        ctx->timing.coder_bytes = ptm.coder_bytes + total; ctx->timing.predict_bytes = ptm.predict_bytes;
        ctx->timing.n_coder_launches = ptm.n_coder_launches; ctx->timing.n_slot_launches = ptm.n_slot_launches;
    }
    if (evp) {
        if (!two) { ctx->timing.generic_ms = elapsed_ev(evp, W3_EV_PREDICT); ctx->timing.pack_ms = d_out ? elapsed_ev(evp, W3_EV_PACK) : 0.f; ctx->timing.total_ms = elapsed_ev(evp, W3_EV_TOTAL); }
        else collect_timing(evp, J.tp, ps.n_apm > 0, ps.has_slot, d_out != nullptr, ctx->timing);
    }
    ctx->timing.n_parts = 1;
    if (d_out && total > out_cap) { ctx->err = "out_cap too small"; return W3_E_NOSPACE; }
    return W3_OK;
}

extern "C" int w3_encode_blocks_device(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *d_in, size_t n, size_t block_size,
                                       uint8_t *d_out, size_t out_cap, uint32_t *d_block_lens, uint64_t *d_total, void *stream) {
    if (!ctx) return W3_E_INVALID;
    if (n && !d_out) return W3_E_INVALID;
    int rc = jobs_idle(ctx);
    if (rc) return rc;
    return encode_core(ctx, jobref(ctx, 0), spec, d_in, n, block_size, d_out, out_cap, d_block_lens, d_total, stream);
}

// ---------------------------------------------------------------------------
// submit / wait: up to two encodes in flight on one context (main.rs:103-109 run for call k+1's predict phase while call k is
// still being coded).  Each job has its own workspace; the predict phases follow each other on one stream, the APM + coder +
// pack stages on another, so in steady state the chip always holds one call's predict kernels and the previous call's APM
// or coder kernel — in the half-CU shapes (w3_predict.h, w3_coder5.h) that let the two share every CU.
// ---------------------------------------------------------------------------
// twophase_predict_b's choice for the slot-state leaves, as far as the spec and the shape decide it (w3_twophase.h)
static bool slot_sorted_by_default(const ParsedSpec &ps, uint32_t nb, size_t block_size, size_t n) {
    if (nb >= W3_SLOT_SORTED_MAX_BLOCKS || block_size > (1ull << 31)) return false;
    size_t n_slot = 0;
    for (int l = 0; l < ps.n_leaves; l++) {
        if (ps.leaf[l].kind != W3_NODE_SLOT_STATE) continue;
        if (ps.leaf[l].log_cells > 16) return false;
        n_slot++;
    }
    // (two jobs in flight hold two sets of event records, 32 bytes per input byte and leaf: beyond 48 GB a set the call stays synchronous —
    // twophase_predict_b then still picks the replay if the records fit beside everything else, or k_slot)
    return 32ull * n_slot * n <= (48ull << 30);
}

// How submitted calls of this spec and size are kept in flight (w3_encode_submit and w3_encode_max_in_flight must agree).
//   ordered pair   step k's coder beside step k+1's rank kernels, APM stage in between: large inputs of models with
//                  wide (sorted) leaves AND an APM stage — the bench model: 14,1xx MiB/s against 13,570 free-running;
//   free-running   every code stage on its own stream as soon as it is submitted: small and medium inputs of any model, and large inputs
//                  of models WITHOUT rank kernels to put the coder beside (1e9 B: Order0 38,099 -> 52,642 MiB/s with three in flight,
//                  main.rs's default model 15,578 -> 17,566; order012, wide leaves but no APM stage, 16,038 -> 16,313 with two).
struct PipelinePlan { bool free_run; int depth; };
static PipelinePlan pipeline_plan(const ParsedSpec &ps, uint32_t nb, uint32_t tune) {
    int n_wide = 0;
    for (int l = 0; l < ps.n_leaves; l++) { const int c = leaf_class(ps.leaf[l]); n_wide += c == LEAF_WIDE1 || c == LEAF_WIDE2 || c == LEAF_WAVE; }
    PipelinePlan p;
    p.free_run = (nb <= W3_FREE_RUN_BLOCKS || n_wide == 0 || ps.n_apm == 0 || (tune & 8192u)) && !(tune & 4096u);   // (W3_OPT_TUNE bit 12: ordered, 13: free-running, whatever the size)
    if (ps.has_slot) p.depth = 2;                       // (event records: 32 bytes per input byte and leaf)
    else if (!p.free_run) p.depth = 2;
    else if (nb <= W3_FREE_RUN4_BLOCKS) p.depth = W3_MAX_JOBS;
    else if (nb <= W3_FREE_RUN_BLOCKS) p.depth = 3;
    else p.depth = n_wide == 0 ? 3 : 2;                 // large inputs: a workspace is 16 bytes per input byte and live leaf (+ 40 per wide leaf)
    return p;
}

// Does w3_encode_submit only enqueue this call (true), or run it to completion inside the call (false: specs outside the predict kernels,
// and specs whose slot-state leaves walk hash maps in HBM)?
static bool submit_pipelines(const w3_ctx *ctx, const ParsedSpec &ps, uint32_t nb, size_t block_size, size_t n) {
    const bool two = nb > 0 && twophase_supported(ps, block_size, n) && ctx->opt_path != W3_PATH_GENERIC;
    // Specs with slot-state leaves are pipelined when the leaves run as the sorted replay (w3_slot2.h: no hash maps sized from the memory
    // that happens to be free) — two jobs at most: a job's event records are 32 bytes per input byte and leaf.
    const bool slot_async = ps.has_slot && slot_sorted_by_default(ps, nb, block_size, n) && !(ctx->tp.variant & (W3_VAR_SLOT_TABLE | W3_VAR_NO_LDS_ATOMICS)) && ctx->tp.lds_order != 0;
    return two && !(ps.has_slot && !slot_async);
}

static int ensure_pipeline(w3_ctx *ctx) {
    // The two stages must not share a hardware queue (HIP maps streams onto GPU_MAX_HW_QUEUES = 4 queues per priority level by
    // default, round-robin: two streams of one level can land on the same queue and then run one after the other).  Streams of
    // different PRIORITY levels use different queues: the code stage — it carries the latency chain — gets the high level.
    if (!ctx->s_pred || !ctx->s_code[W3_MAX_JOBS - 1]) {
        int lo_p = 0, hi_p = 0;
        HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
        const bool pred_high = (ctx->tp.tune & 1u) != 0;   // W3_OPT_TUNE bit 0
        const bool plain = !(ctx->tp.tune & (1u | 16u));   // (tuning variants create their own)
        if (!ctx->s_pred && plain) ctx->s_pred = stream_pool().take(ctx->device * 8L);
        if (!ctx->s_pred) HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->s_pred, hipStreamNonBlocking, pred_high ? hi_p : 0));
        // (created one after the other on one level: HIP deals that level's hardware queues out round-robin, so the W3_MAX_JOBS = 4 code
        // streams get a queue each and the free-running jobs' coders really run side by side)
        for (auto &sc : ctx->s_code) {
            if (!sc && plain) sc = stream_pool().take(ctx->device * 8L + 1);
            if (!sc) HIPCHK(ctx, hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, (pred_high || (ctx->tp.tune & 16u)) ? 0 : hi_p));
        }
        ctx->pooled_streams = plain;
    }
    for (int j = 0; j < W3_MAX_JOBS; j++) {
        JobState &st = ctx->js[j];
        if (!st.ev_done) HIPCHK(ctx, hipEventCreateWithFlags(&st.ev_done, hipEventDisableTiming));
        if (!st.ev_in) HIPCHK(ctx, hipEventCreateWithFlags(&st.ev_in, hipEventDisableTiming));
        if (!st.ev_a) HIPCHK(ctx, hipEventCreateWithFlags(&st.ev_a, hipEventDisableTiming));
        if (!st.ev_apm) HIPCHK(ctx, hipEventCreateWithFlags(&st.ev_apm, hipEventDisableTiming));
        if (!st.h_status) HIPCHK(ctx, hipHostMalloc((void **)&st.h_status, 32, hipHostMallocDefault));
    }
    for (auto &x : ctx->jx)
        for (auto &e : x.ev)
            if (!e) HIPCHK(ctx, hipEventCreate(&e));
    return W3_OK;
}

// APM stages + coder + pack + status read-back of an enqueued job, on the code stream
static int enqueue_code(w3_ctx *ctx, JobRef &J, hipEvent_t wait_ev, hipEvent_t rec_after_apm) {
    JobState &st = J.st;
    hipStream_t sp = ctx->s_pred, sc = st.sc;
    hipEvent_t *evp = st.timed ? J.ev : nullptr;
    int rc = tp_code_stage(J.tp, sp, sc, st.ps, st.d_in, st.n, st.block_size, st.nb, (uint8_t *)J.stripes.p, st.cap, st.d_block_lens, (uint32_t *)J.flag.p,
                           wait_ev, rec_after_apm, evp, &st.tm, ctx->err);
    if (rc) return rc;
    if (evp) HIPCHK(ctx, hipEventRecord(evp[2 * W3_EV_PACK], sc));
    rc = run_pack(ctx, J, sc, (const uint8_t *)J.stripes.p, st.cap, st.d_block_lens, st.nb, st.d_out, st.out_cap, st.d_total);
    if (evp) { HIPCHK(ctx, hipEventRecord(evp[2 * W3_EV_PACK + 1], sc)); HIPCHK(ctx, hipEventRecord(evp[2 * W3_EV_TOTAL + 1], sc)); }
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(st.h_status, J.flag.p, 16, hipMemcpyDeviceToHost, sc));
    HIPCHK(ctx, hipMemcpyAsync(st.h_status + 4, st.d_total, 8, hipMemcpyDeviceToHost, sc));
    HIPCHK(ctx, hipEventRecord(st.ev_done, sc));
    st.code_pending = false;
    return W3_OK;
}

extern "C" int w3_encode_submit(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *d_in, size_t n, size_t block_size,
                                uint8_t *d_out, size_t out_cap, uint32_t *d_block_lens, uint64_t *d_total, void *stream, int *job) {
    if (!ctx || !job) return W3_E_INVALID;
    *job = -1;
    int rc = check_args(ctx, n, block_size);
    if (rc) return rc;
    if (n && (!d_out || !d_in || !d_block_lens)) return W3_E_INVALID;
    if (!d_total) { ctx->err = "w3_encode_submit needs d_total"; return W3_E_INVALID; }
    ParsedSpec ps;
    if ((rc = parse_spec(spec, ps))) { ctx->err = "malformed model spec"; return rc; }
    const uint32_t nb = (uint32_t)((n + block_size - 1) / block_size);
    const PipelinePlan plan = pipeline_plan(ps, nb, ctx->tp.tune);
    const bool free_run = plan.free_run;
    const int depth = plan.depth;
    int in_flight = 0;
    for (const auto &o : ctx->js) in_flight += o.state != 0;
    int j = ctx->next_job % depth;
    if (ctx->js[j].state != 0) {   // (a free slot further on: jobs may be waited for in any order)
        for (int k = 0; k < depth; k++)
            if (ctx->js[k].state == 0) { j = k; break; }
    }
    if (in_flight >= depth || ctx->js[j].state != 0) {
        ctx->err = std::to_string(in_flight) + " jobs are in flight already (at most " + std::to_string(depth) + " for an input of this size): w3_encode_wait the oldest one first";
        return W3_E_INVALID;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!submit_pipelines(ctx, ps, nb, block_size, n)) {
        // Not pipelined: the lane-per-block kernels (any spec the predict kernels do not cover) and specs whose slot-state leaves walk
        // hash maps in HBM (k_slot: sized from the memory that is free at the time) run to completion here, on job 0's workspace.
        for (int k = 0; k < W3_MAX_JOBS; k++) {   // let the other jobs' kernels finish first; their status is in pinned memory already
            if (ctx->js[k].state != 1) continue;
            if (ctx->js[k].code_pending) { JobRef O = jobref(ctx, k); if ((rc = enqueue_code(ctx, O, nullptr, nullptr))) return rc; }
            HIPCHK(ctx, hipEventSynchronize(ctx->js[k].ev_done));
            if (k == 0 && ctx->js[0].timed && !ctx->js[0].tm_snap) {   // job 0's events are about to be recorded again
                JobRef O = jobref(ctx, 0);
                memset(&O.st.tm_ev, 0, sizeof O.st.tm_ev);
                collect_timing(O.ev, O.tp, O.st.has_apm, O.st.has_slot, true, O.st.tm_ev);
                O.st.tm_snap = true;
            }
        }
        // Completed by w3_encode_wait like any other job: THAT call returns what the synchronous call returned (W3_E_NOSPACE with
        // d_total = the need included), so a caller that follows the submit / wait contract sees one behaviour for every spec.
        const w3_timing keep = ctx->timing;
        ctx->js[j].sync_rc = encode_core(ctx, jobref(ctx, 0), spec, d_in, n, block_size, d_out, out_cap, d_block_lens, d_total, stream);
        ctx->js[j].state = 2; ctx->js[j].tm = ctx->timing;
        ctx->timing = keep;
        *job = j; ctx->next_job = j + 1; ctx->last_job = j;
        return W3_OK;
    }
    if ((rc = ensure_pipeline(ctx))) return rc;
    JobRef J = jobref(ctx, j);
    JobState &st = J.st;
    // keep the call (the spec and its HuffHistory tables are the caller's memory)
    st.spec = *spec;
    if (ps.n_huff) { memcpy(st.huff_copy, ps.huff, sizeof(w3_huff_table) * ps.n_huff); st.spec.huff = st.huff_copy; ps.huff = st.huff_copy; }
    st.ps = ps;
    st.d_in = d_in; st.n = n; st.block_size = block_size; st.d_out = d_out; st.out_cap = out_cap; st.d_block_lens = d_block_lens; st.d_total = d_total;
    st.has_apm = ps.n_apm > 0; st.has_slot = ps.has_slot; st.timed = ctx->opt_timing != 0; st.tm_snap = false;
    st.nb = nb;
    st.sc = free_run ? ctx->s_code[j] : ctx->s_code[0];
    memset(&st.tm, 0, sizeof st.tm);
    hipStream_t sp = ctx->s_pred;
    // after whatever produced d_in on the caller's stream
    hipStream_t s_in = stream ? (hipStream_t)stream : ctx->stream;
    HIPCHK(ctx, hipEventRecord(st.ev_in, s_in));
    HIPCHK(ctx, hipStreamWaitEvent(sp, st.ev_in, 0));
    ENSURE(ctx, J.flag, 16);
    ENSURE(ctx, J.bits, (size_t)nb * 4);
    if ((rc = stage_huff_job(ctx, J, sp, ps))) return rc;
    if (ps.is_cm()) {
        CmArgs lut;
        if ((rc = cm_luts(ctx, sp, lut))) return rc;
        ctx->tp.stretch = lut.stretch; ctx->tp.squash = lut.squash; ctx->tp.st = lut.st;
    }
    sync_job_options(ctx, J);
    if ((rc = attach_aux_streams(ctx, J, j))) return rc;
    J.tp.half_cu = !(ctx->tp.variant & W3_VAR_FULL_CU);
    J.tp.verify_calls = ctx->n_encodes++;   // (the sample's rotation is the context's: a call's job slot does not matter)
    st.cap = default_stripe_cap(block_size);
    ENSURE(ctx, J.stripes, (size_t)nb * st.cap);
    hipEvent_t *evp = st.timed ? J.ev : nullptr;
    HIPCHK(ctx, hipMemsetAsync(J.flag.p, 0, 16, sp));
    if (evp) HIPCHK(ctx, hipEventRecord(evp[2 * W3_EV_TOTAL], sp));
    J.tp.out_bits = (uint32_t *)J.bits.p;
    J.tp.order_fault = (uint32_t *)J.flag.p + 2;
#ifdef W3_TUNING
    J.tp.apm_oob = (uint32_t *)J.flag.p + 3;
#endif
    // The order in which the two jobs' kernels reach the chip (measured, profiles/r3_pipeline/: kernels that fill the LDS — the
    // partition passes, the time-ordered leaves, k_apm0 — only slow each other down when they share CUs; the rank kernels, bound
    // by their scattered stores, and the coder, one latency chain per lane, run well side by side):
    //     first predict half of THIS job  ->  APM stages of the OTHER job  ->  coder of the other job  BESIDE  rank kernels of this job
    // So the other job's code stage is enqueued here, between this job's two predict halves (W3_OPT_TUNE bit 2: no such order).
    // Free-running jobs (small inputs) need no such order: the chip is mostly idle while a call is coded, so every job's code stage
    // goes to its own stream at once and the coders of up to W3_MAX_JOBS calls run side by side (each a latency chain on a few CUs).
    const bool ordered = !(ctx->tp.tune & 4u) && !free_run;
    const int prev = ctx->last_job >= 0 && ctx->last_job != j && ctx->js[ctx->last_job].state == 1 && ctx->js[ctx->last_job].code_pending ? ctx->last_job : -1;
    if (prev >= 0 && !ordered) {   // (an ordered job before a free-running one: its code stage goes out now)
        JobRef O = jobref(ctx, prev);
        if ((rc = enqueue_code(ctx, O, nullptr, nullptr))) return rc;
    }
    rc = twophase_predict_a(J.tp, sp, ps, d_in, n, block_size, nb, evp, ctx->err, ordered);
    if (!rc && ordered) {
        HIPCHK(ctx, hipEventRecord(st.ev_a, sp));
        if (prev >= 0) {
            JobRef O = jobref(ctx, prev);
            rc = enqueue_code(ctx, O, st.ev_a, O.st.ev_apm);
            if (!rc) HIPCHK(ctx, hipStreamWaitEvent(sp, O.st.ev_apm, 0));
        }
    }
    if (!rc) rc = twophase_predict_b(J.tp, sp, ps, d_in, n, block_size, nb, tp_plan(J.tp, ps).need_P, nullptr, evp, &st.tm, ctx->err);
    if (!rc) rc = tp_after_predict(J.tp, sp, ps, d_in, n, block_size, nb, (uint32_t *)J.flag.p, ctx->err);
    if (!rc) {
        st.state = 1; st.code_pending = true;
        if (!ordered) rc = enqueue_code(ctx, J, nullptr, nullptr);
    }
    if (rc) {   // leave nothing of either job running behind an error return
        (void)hipDeviceSynchronize();
        st.state = 0; st.code_pending = false;
        return rc;
    }
    *job = j; ctx->next_job = j + 1; ctx->last_job = j;
    return W3_OK;
}

extern "C" int w3_encode_max_in_flight(const w3_model_spec *spec, size_t n, size_t block_size) {
    if (!block_size) return 0;
    const size_t nb = (n + block_size - 1) / block_size;
    ParsedSpec ps;   // (no spec: a model with wide leaves and an APM stage, the most conservative answer)
    if (spec) { if (parse_spec(spec, ps)) return 0; }
    else { ps.n_leaves = 1; ps.leaf[0] = w3_node{}; ps.leaf[0].kind = W3_NODE_ORDERN; ps.leaf[0].bits = 19; ps.leaf[0].align = 3; ps.n_apm = 1; }
    return pipeline_plan(ps, (uint32_t)std::min<size_t>(nb, 0xFFFFFFFFu), 0u).depth;
}

extern "C" int w3_encode_wait(w3_ctx *ctx, int job) {
    if (!ctx || job < 0 || job >= W3_MAX_JOBS) return W3_E_INVALID;
    JobRef J = jobref(ctx, job);
    JobState &st = J.st;
    if (st.state == 0) { ctx->err = "no such job in flight"; return W3_E_INVALID; }
    st.total_valid = false;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (st.state == 2) { st.state = 0; ctx->timing = st.tm; return st.sync_rc; }
    if (st.code_pending) {   // no later submit has placed this job's code stage: it goes out now
        const int rc = enqueue_code(ctx, J, nullptr, nullptr);
        if (rc) { (void)hipDeviceSynchronize(); st.state = 0; st.code_pending = false; return rc; }
    }
    HIPCHK(ctx, hipEventSynchronize(st.ev_done));
    st.state = 0;
    const uint32_t f0 = st.h_status[0], redo = st.h_status[1], mism = st.h_status[2];
    uint64_t total;
    memcpy(&total, st.h_status + 4, 8);
    if ((f0 & 2u) && !mism) { ctx->err = "coder pipeline timeout (internal error)"; return W3_E_HIP; }
    if (st.h_status[3]) { ctx->err = "APM kernel: stores outside the stage's stream and the sink (W3_TUNING store guard)"; return W3_E_HIP; }
    if (f0 || redo || mism) {
        // Rare: a stripe overflowed the 2N+64 bound, the fast coder handed blocks back, or the sampled verification saw the LDS-add
        // rounds misbehave.  Let the other job's kernels finish (its output is complete then, its status in pinned memory) and
        // run this call again synchronously on this job's workspace: encode_core's own retry loop deals with each case.
        HIPCHK(ctx, hipDeviceSynchronize());
        if (mism) {
            ctx->tp.variant |= W3_VAR_NO_LDS_ATOMICS; ctx->tp.lds_order = 0;
            for (auto &x : ctx->jx) { x.tp.variant |= W3_VAR_NO_LDS_ATOMICS; x.tp.lds_order = 0; }
        }
        const int rc = encode_core(ctx, J, &st.spec, st.d_in, st.n, st.block_size, st.d_out, st.out_cap, st.d_block_lens, st.d_total, ctx->stream);
        ctx->timing.n_lds_faults += mism;
        return rc;
    }
    st.total_out = total; st.total_valid = true;
    memset(&ctx->timing, 0, sizeof ctx->timing);
    ctx->timing.path = W3_PATH_TWOPHASE;
    ctx->timing.coder_bytes = st.tm.coder_bytes + total; ctx->timing.predict_bytes = st.tm.predict_bytes;
    ctx->timing.n_coder_launches = st.tm.n_coder_launches; ctx->timing.n_slot_launches = st.tm.n_slot_launches;
    ctx->timing.n_parts = 1;
    if (st.timed && st.tm_snap) {
        const w3_timing &e = st.tm_ev;
        ctx->timing.predict_ms = e.predict_ms; ctx->timing.coder_ms = e.coder_ms; ctx->timing.apm_ms = e.apm_ms; ctx->timing.slot_ms = e.slot_ms;
        ctx->timing.achash_ms = e.achash_ms; ctx->timing.n_wide = e.n_wide; ctx->timing.small_ms = e.small_ms; ctx->timing.pack_ms = e.pack_ms; ctx->timing.total_ms = e.total_ms;
        memcpy(ctx->timing.part_ms, e.part_ms, sizeof e.part_ms); memcpy(ctx->timing.rank_ms, e.rank_ms, sizeof e.rank_ms);
    } else if (st.timed) collect_timing(J.ev, J.tp, st.has_apm, st.has_slot, true, ctx->timing);
    if (total > st.out_cap) { ctx->err = "out_cap too small"; return W3_E_NOSPACE; }
    return W3_OK;
}

// ---------------------------------------------------------------------------
// Host buffers, asynchronous (ABI v8): compress() of main.rs:89-113 reads a file and writes a file, so what a host sees is
// PCIe in, encode, PCIe out.  A call here is those three as a pipeline ACROSS calls: the input of call k+1 travels while call k
// is encoded (w3_encode_submit: up to four encodes in flight) and the streams of call k-1 travel back.  Copies run on two
// streams of their own (one per direction: the link is full duplex); from pinned host memory they are asynchronous, from
// pageable memory HIP stages them and the enqueueing call blocks — the encodes already submitted keep the GPU busy meanwhile.
// ---------------------------------------------------------------------------
static int host_streams(w3_ctx *ctx) {
    // The copies of BOTH directions go to the context's own stream — idle while calls are pipelined (only the synchronous entry points
    // launch on it) — in the order the host needs them: a call's input long before its encode is submitted, a call's streams after it
    // has been waited for; together 24 ms of a 68 ms step at 1e9 B, so the link need not run full duplex.  NO stream is created: HIP maps
    // streams onto 4 hardware queues per priority level (unless GPU_MAX_HW_QUEUES says otherwise) by the queues' use at creation, and a
    // copy stream that lands on the predict stream's queue holds that call's kernels back for the length of a copy (measured: 84.5 ms
    // per call with two copy streams of their own against 72.6 with 8 hardware queues in the environment, profiles/r4_host_path/).
    // W3_OPT_TUNE bit 16: two copy streams of their own, one per direction (for hosts that do raise GPU_MAX_HW_QUEUES).
    if (ctx->tp.tune & 65536u) {
        if (!ctx->s_h2d_own) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_h2d_own, hipStreamNonBlocking));
        if (!ctx->s_d2h_own) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_d2h_own, hipStreamNonBlocking));
        ctx->s_h2d = ctx->s_h2d_own; ctx->s_d2h = ctx->s_d2h_own;
    } else {
        ctx->s_h2d = ctx->s_d2h = ctx->stream;
    }
    for (auto &h : ctx->hj)
        if (!h.ev_d2h) HIPCHK(ctx, hipEventCreateWithFlags(&h.ev_d2h, hipEventDisableTiming));
    return W3_OK;
}

static int host_depth(const w3_model_spec *spec, size_t n, size_t block_size) {
    return std::min(W3_MAX_HOST_JOBS, w3_encode_max_in_flight(spec, n, block_size) + 1);
}

// Submit the encodes of host jobs whose input is on its way, oldest first, while device job slots are free.
static void host_start_pending(w3_ctx *ctx) {
    for (;;) {
        HostJob *h = nullptr;
        for (auto &x : ctx->hj)
            if (x.state == 1 && (!h || x.seq < h->seq)) h = &x;
        if (!h) return;
        int in_flight = 0;
        for (const auto &o : ctx->js) in_flight += o.state != 0;
        if (in_flight >= w3_encode_max_in_flight(&h->spec, h->n, h->block_size)) return;
        // (the job starts behind everything enqueued on the H2D stream so far: its own input was the last of it)
        h->rc = w3_encode_submit(ctx, &h->spec, (const uint8_t *)h->d_in.p, h->n, h->block_size, (uint8_t *)h->d_out.p, h->dcap,
                                 (uint32_t *)h->d_lens.p, (uint64_t *)h->d_total.p, ctx->s_h2d, &h->djob);
        h->state = h->rc ? 3 : 2;   // (a refused submit is reported by the wait)
    }
}

// state 2 -> 3: wait for the job's encode; rc and the compressed size are known afterwards
static void host_finish_device(w3_ctx *ctx, HostJob &h) {
    h.rc = w3_encode_wait(ctx, h.djob);
    h.tm = ctx->timing;
    h.total = 0;
    h.state = 3;
    auto read_total = [&]() -> int {
        HIPCHK(ctx, hipMemcpyAsync(&h.total, h.d_total.p, 8, hipMemcpyDeviceToHost, ctx->s_d2h));
        HIPCHK(ctx, hipStreamSynchronize(ctx->s_d2h));
        return W3_OK;
    };
    if (h.rc == W3_OK || h.rc == W3_E_NOSPACE) {
        const JobState &ds = ctx->js[h.djob];
        if (ds.total_valid) h.total = ds.total_out;   // (the usual case: no device access, no wait for a copy that is in flight on the copy stream)
        else { const int r = read_total(); if (r) h.rc = r; }
    }
    if (h.rc == W3_E_NOSPACE && h.total > h.dcap) {
        // The device buffer is sized for the realistic bound (2 n + 64 per block: the stripes' own); a call beyond it (adversarial
        // input: up to 16 n) is encoded again, alone, with the room it asked for.
        auto redo = [&]() -> int {
            HIPCHK(ctx, hipDeviceSynchronize());   // (the other jobs' kernels: their outputs are complete afterwards, their status words in pinned memory)
            ENSURE(ctx, h.d_out, (size_t)h.total);
            h.dcap = (size_t)h.total;
            ParsedSpec ps;
            int rc = parse_spec(&h.spec, ps);
            if (rc) return rc;
            const int slot = submit_pipelines(ctx, ps, (uint32_t)h.nb, h.block_size, h.n) ? h.djob : 0;
            rc = encode_core(ctx, jobref(ctx, slot), &h.spec, (const uint8_t *)h.d_in.p, h.n, h.block_size, (uint8_t *)h.d_out.p, h.dcap,
                             (uint32_t *)h.d_lens.p, (uint64_t *)h.d_total.p, ctx->stream);
            if (rc) return rc;
            return read_total();
        };
        h.rc = redo();
    }
}

static int host_submit_core(w3_ctx *ctx, const w3_model_spec *spec, const ParsedSpec &ps, const uint8_t *in, size_t n, size_t block_size,
                            uint8_t *out, size_t out_cap, uint32_t *block_lens, int *hjob) {
    int rc = host_streams(ctx);
    if (rc) return rc;
    const size_t nb = (n + block_size - 1) / block_size;
    int busy = 0, slot = -1;
    for (int k = 0; k < W3_MAX_HOST_JOBS; k++) {
        if (ctx->hj[k].state != 0) busy++;
        else if (slot < 0) slot = k;
    }
    const int depth = host_depth(spec, n, block_size);
    if (busy >= depth || slot < 0) {
        ctx->err = std::to_string(busy) + " host-buffer jobs are in flight already (at most " + std::to_string(depth) + " for an input of this size): w3_encode_host_wait the oldest one first";
        return W3_E_INVALID;
    }
    HostJob &h = ctx->hj[slot];
    h.spec = *spec;   // (the spec and its HuffHistory tables are the caller's memory)
    if (ps.n_huff) { memcpy(h.huff_copy, ps.huff, sizeof(w3_huff_table) * ps.n_huff); h.spec.huff = h.huff_copy; }
    h.n = n; h.block_size = block_size; h.nb = nb; h.out = out; h.out_cap = out_cap; h.block_lens = block_lens;
    h.djob = -1; h.rc = W3_OK; h.total = 0;
    memset(&h.tm, 0, sizeof h.tm);
    // the device output buffer: the realistic bound, never more than the hard one (a call beyond it is redone: host_finish_device)
    h.dcap = std::min<size_t>(w3_max_compressed_size(n, block_size), 2 * n + 64 * nb + 64);
    ENSURE(ctx, h.d_in, std::max<size_t>(n, 16));
    ENSURE(ctx, h.d_out, std::max<size_t>(h.dcap, 16));
    ENSURE(ctx, h.d_lens, std::max<size_t>(nb * 4, 16));
    ENSURE(ctx, h.d_total, 8);
    HIPCHK(ctx, hipMemcpyAsync(h.d_in.p, in, n, hipMemcpyHostToDevice, ctx->s_h2d));
    h.seq = ++ctx->hseq;
    h.state = 1;
    host_start_pending(ctx);
    *hjob = slot;
    return W3_OK;
}

// out / out_cap: where the streams go (w3_encode_blocks binds a chunk's destination only now: it is the sum of the earlier chunks'
// sizes).  *out_len is set even on W3_E_NOSPACE; the length table is copied in either case.
static int host_wait_core(w3_ctx *ctx, int hjob, uint8_t *out, size_t out_cap, size_t *out_len) {
    HostJob &h = ctx->hj[hjob];
    // jobs are encoded in the order they were submitted: whatever is older goes through the device first
    while (h.state == 1 || h.state == 2) {
        HostJob *o = nullptr;
        for (auto &x : ctx->hj)
            if (x.state == 2 && (!o || x.seq < o->seq)) o = &x;
        if (o) host_finish_device(ctx, *o);
        const int before = h.state;
        host_start_pending(ctx);   // FIRST: the next call's kernels reach the GPU before this call's streams start travelling back
        if (!o && h.state == before) { h.state = 0; ctx->err = "host-buffer job could not be submitted (internal error)"; return W3_E_HIP; }
    }
    int rc = h.rc;
    *out_len = (size_t)h.total;
    if (rc == W3_OK || rc == W3_E_NOSPACE) {
        auto fetch = [&]() -> int {
            if (h.nb && h.block_lens) HIPCHK(ctx, hipMemcpyAsync(h.block_lens, h.d_lens.p, h.nb * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
            const bool fits = rc == W3_OK && h.total <= out_cap && (out || !h.total);
            if (fits && h.total) HIPCHK(ctx, hipMemcpyAsync(out, h.d_out.p, (size_t)h.total, hipMemcpyDeviceToHost, ctx->s_d2h));
            HIPCHK(ctx, hipEventRecord(h.ev_d2h, ctx->s_d2h));
            HIPCHK(ctx, hipEventSynchronize(h.ev_d2h));
            if (!fits) { if (rc == W3_OK) ctx->err = "out_cap too small"; return W3_E_NOSPACE; }
            return W3_OK;
        };
        rc = fetch();
    }
    ctx->timing = h.tm;
    h.state = 0;
    return rc;
}

extern "C" int w3_encode_host_max_in_flight(const w3_model_spec *spec, size_t n, size_t block_size) {
    if (!block_size || (spec && w3_spec_validate(spec))) return 0;
    return host_depth(spec, n, block_size);
}

extern "C" int w3_encode_host_submit(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *in, size_t n, size_t block_size,
                                     uint8_t *out, size_t out_cap, uint32_t *block_lens, int *hjob) {
    if (!ctx || !hjob) return W3_E_INVALID;
    *hjob = -1;
    int rc = check_args(ctx, n, block_size);
    if (rc) return rc;
    ParsedSpec ps;
    if ((rc = parse_spec(spec, ps))) { ctx->err = "malformed model spec"; return rc; }
    if (n == 0 || !in || !block_lens) { ctx->err = "w3_encode_host_submit needs input and a length table"; return W3_E_INVALID; }
    for (const auto &st : ctx->js)   // (the two levels are not mixed: the host jobs count the device job slots as theirs)
        if (st.state != 0) {
            bool ours = false;
            for (const auto &h : ctx->hj) ours |= h.state == 2 && &ctx->js[h.djob] == &st;
            if (!ours) { ctx->err = "w3_encode_submit jobs are in flight on this context: w3_encode_wait them first"; return W3_E_INVALID; }
        }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return host_submit_core(ctx, spec, ps, in, n, block_size, out, out_cap, block_lens, hjob);
}

extern "C" int w3_encode_host_wait(w3_ctx *ctx, int hjob, size_t *out_len) {
    if (!ctx || !out_len || hjob < 0 || hjob >= W3_MAX_HOST_JOBS) return W3_E_INVALID;
    *out_len = 0;
    if (ctx->hj[hjob].state == 0) { ctx->err = "no such host-buffer job in flight"; return W3_E_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return host_wait_core(ctx, hjob, ctx->hj[hjob].out, ctx->hj[hjob].out_cap, out_len);
}

// ---------------------------------------------------------------------------
// ACStats (helpers.rs:60-90): the counting sink every published figure of the reference comes from
// ---------------------------------------------------------------------------
extern "C" int w3_encode_stats_device(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *d_in, size_t n, size_t block_size,
                                      uint32_t *d_block_bits, void *stream) {
    int rc = check_args(ctx, n, block_size);
    if (rc) return rc;
    const size_t nb = (n + block_size - 1) / block_size;
    if (nb == 0) return w3_spec_validate(spec);
    if (!d_in || !d_block_bits) return W3_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = jobs_idle(ctx))) return rc;
    ENSURE(ctx, ctx->lens, nb * 4);
    rc = encode_core(ctx, jobref(ctx, 0), spec, d_in, n, block_size, nullptr, 0, (uint32_t *)ctx->lens.p, nullptr, stream);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(d_block_bits, ctx->bits.p, nb * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return W3_OK;
}

extern "C" int w3_encode_stats(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *in, size_t n, size_t block_size, uint32_t *block_bits) {
    int rc = check_args(ctx, n, block_size);
    if (rc) return rc;
    const size_t nb = (n + block_size - 1) / block_size;
    if (nb == 0) return w3_spec_validate(spec);
    if (!in || !block_bits) return W3_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = jobs_idle(ctx))) return rc;
    ENSURE(ctx, ctx->io_in, n);
    ENSURE(ctx, ctx->lens, nb * 4);
    HIPCHK(ctx, hipMemcpyAsync(ctx->io_in.p, in, n, hipMemcpyHostToDevice, ctx->stream));
    rc = encode_core(ctx, jobref(ctx, 0), spec, (const uint8_t *)ctx->io_in.p, n, block_size, nullptr, 0, (uint32_t *)ctx->lens.p, nullptr, ctx->stream);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy(block_bits, ctx->bits.p, nb * 4, hipMemcpyDeviceToHost));
    return W3_OK;
}

extern "C" int w3_decode_blocks_device(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *d_in, size_t in_len, const uint32_t *d_block_lens,
                                       size_t nblocks, size_t block_size, uint64_t orig_len, uint8_t *d_out, void *stream) {
    int rc = check_args(ctx, (size_t)orig_len, block_size);
    if (rc) return rc;
    if ((rc = jobs_idle(ctx))) return rc;   // (job 0's HuffHistory tables, length scan and model tables)
    ParsedSpec ps;
    if ((rc = parse_spec(spec, ps))) { ctx->err = "malformed model spec"; return rc; }
    const uint64_t nb = (orig_len + block_size - 1) / block_size;
    if (nb != nblocks) { ctx->err = "nblocks does not match orig_len/block_size"; return W3_E_INVALID; }
    if (nb == 0) return W3_OK;
    if (!d_in || !d_block_lens || !d_out) return W3_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if ((rc = stage_huff(ctx, s, ps))) return rc;
    {   // the length table must not claim more than the caller's buffer holds: the kernels read cin + offset for clens[b] bytes
        ENSURE(ctx, ctx->coffs, (size_t)nb * 8);
        ENSURE(ctx, ctx->total, 8);
        hipLaunchKernelGGL(k_scan_lens, dim3(1), dim3(1024), 0, s, d_block_lens, (uint64_t *)ctx->coffs.p, (uint64_t *)ctx->total.p, (uint32_t)nb);
        uint64_t total = 0;
        HIPCHK(ctx, hipMemcpyAsync(&total, ctx->total.p, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        if (total > in_len) { ctx->err = "block length table claims " + std::to_string(total) + " compressed bytes, the buffer holds " + std::to_string(in_len); return W3_E_FORMAT; }
    }
    rc = ps.is_cm() ? cm_decode(ctx, s, ps, d_in, d_block_lens, (uint32_t)nb, block_size, orig_len, d_out)
                    : generic_decode(ctx, s, ps, d_in, d_block_lens, (uint32_t)nb, block_size, orig_len, d_out);
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(s));
    return W3_OK;
}

// ---------------------------------------------------------------------------
// host-buffer entry points
// ---------------------------------------------------------------------------
// How a host-buffer call is cut into pipelined pieces (whole blocks each).  The pieces are calls of w3_encode_host_submit, so piece
// k+1's input travels while piece k is encoded and piece k-1's streams travel back; what a single call cannot hide is its first
// piece's H2D and its last piece's coder chain (8 x block_size dependent steps per lane whatever the block count) and D2H.
//   - calls that w3_encode_submit would run synchronously (lane-per-block specs, slot leaves on hash maps in HBM): one piece — they
//     take hundreds of milliseconds per GB, PCIe is a few percent of that;
//   - up to W3_FREE_RUN4_BLOCKS blocks: one piece (four coders of such pieces overlap, but one call has only one);
//   - beyond: pieces of at most W3_FREE_RUN4_BLOCKS blocks, equal in size — the four-in-flight regime of DESIGN.md 2.8
//     (measured at 1e9 B from pinned memory, tools/host_api_rate.py: DESIGN.md section 5).
// W3_OPT_HOST_CHUNK_BLOCKS overrides the piece size (tests: ragged pieces; measurements).
// blocks per device call of a host-buffer entry point whose input exceeds what one device call handles (check_args): 2 GiB worth
static size_t host_call_cap_blocks(size_t block_size) { return std::max<size_t>(1, ((size_t)1 << 31) / block_size); }

static size_t host_chunk_blocks(const w3_ctx *ctx, const ParsedSpec &ps, size_t nb, size_t block_size, size_t n) {
    const size_t cap = host_call_cap_blocks(block_size);   // (a host buffer of any length goes through in pieces, as the reference streams any length: main.rs:97-109)
    if (!submit_pipelines(ctx, ps, (uint32_t)std::min<size_t>(nb, cap), block_size, std::min(n, cap * block_size))) return std::min(nb, cap);
    if (ctx->host_chunk_blocks) return std::min<size_t>(nb, ctx->host_chunk_blocks);
    if (nb <= W3_FREE_RUN4_BLOCKS) return nb;
    const size_t pieces = (nb + W3_FREE_RUN4_BLOCKS - 1) / W3_FREE_RUN4_BLOCKS;
    return std::min((nb + pieces - 1) / pieces, cap);
}

extern "C" int w3_encode_blocks(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *in, size_t n, size_t block_size, uint8_t *out,
                                size_t out_cap, size_t *out_len, uint32_t *block_lens) {
    int rc = check_args(ctx, n, block_size, false);   // (any length: the pieces below are the device calls, each under the per-call limit)
    if (rc) return rc;
    if (out_len) *out_len = 0;
    const size_t nb = (n + block_size - 1) / block_size;
    if (nb == 0) return w3_spec_validate(spec);
    if (!in || !block_lens || !out_len) return W3_E_INVALID;
    if ((rc = jobs_idle(ctx))) return rc;
    ParsedSpec ps;
    if ((rc = parse_spec(spec, ps))) { ctx->err = "malformed model spec"; return rc; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t cb = host_chunk_blocks(ctx, ps, nb, block_size, n);
    // The pieces in flight, oldest first.  A piece's streams go to `out` at the sum of the earlier pieces' sizes, known when it is
    // waited for; once the caller's buffer is full the later pieces are still encoded (*out_len must hold the size needed) but
    // only their length tables are fetched.
    int q[W3_MAX_HOST_JOBS], qn = 0;
    size_t off = 0;
    int first_err = W3_OK;
    w3_timing sum;
    memset(&sum, 0, sizeof sum);
    auto wait_oldest = [&]() {
        const int hjob = q[0];
        for (int k = 1; k < qn; k++) q[k - 1] = q[k];
        qn--;
        size_t len = 0;
        const bool room = first_err == W3_OK && out && off <= out_cap;
        const int r = host_wait_core(ctx, hjob, room ? out + off : nullptr, room ? out_cap - off : 0, &len);
        off += len;
        if (r != W3_OK && first_err == W3_OK) first_err = r;
        const w3_timing &t = ctx->timing;
        sum.predict_ms += t.predict_ms; sum.coder_ms += t.coder_ms; sum.pack_ms += t.pack_ms; sum.generic_ms += t.generic_ms; sum.total_ms += t.total_ms;
        sum.apm_ms += t.apm_ms; sum.slot_ms += t.slot_ms; sum.achash_ms += t.achash_ms; sum.small_ms += t.small_ms;
        for (int w = 0; w < 4; w++) { sum.part_ms[w] += t.part_ms[w]; sum.rank_ms[w] += t.rank_ms[w]; }
        sum.path = t.path; sum.n_wide = t.n_wide; sum.n_parts += 1;
        sum.n_coder_launches += t.n_coder_launches; sum.coder_bytes += t.coder_bytes; sum.predict_bytes += t.predict_bytes;
        sum.n_recoded_blocks += t.n_recoded_blocks; sum.n_slot_launches += t.n_slot_launches; sum.n_lds_faults += t.n_lds_faults;
    };
    // Equal pieces.  (A half-size first and last piece — the call's first H2D + predict phase and its last predict + APM overlap nothing —
    // was measured: 97.6 against 96.9 ms at 1e9 B, 91.8 against 53.4 ms at 4e8 B; profiles/r4_host_path/.)
    for (size_t b0 = 0; b0 < nb; b0 += cb) {
        const size_t lo = b0 * block_size, hi = std::min(n, (b0 + cb) * block_size);
        // (an error that is not "out of room" ends the call: nothing more is submitted, what is in flight is drained below)
        if (first_err != W3_OK && first_err != W3_E_NOSPACE) break;
        while (qn >= host_depth(spec, hi - lo, block_size)) wait_oldest();
        if (first_err != W3_OK && first_err != W3_E_NOSPACE) break;
        int hjob = -1;
        rc = host_submit_core(ctx, spec, ps, in + lo, hi - lo, block_size, nullptr, 0, block_lens + b0, &hjob);
        if (rc) { if (first_err == W3_OK) first_err = rc; break; }
        q[qn++] = hjob;
    }
    while (qn) wait_oldest();
    ctx->timing = sum;
    if (first_err != W3_OK && first_err != W3_E_NOSPACE) return first_err;
    *out_len = off;
    if (first_err == W3_E_NOSPACE || off > out_cap || !out) { ctx->err = "out_cap too small"; return W3_E_NOSPACE; }
    return W3_OK;
}

extern "C" int w3_decode_blocks(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *in, size_t in_len, const uint32_t *block_lens, size_t nblocks,
                                size_t block_size, uint64_t orig_len, uint8_t *out) {
    int rc = check_args(ctx, (size_t)orig_len, block_size, false);
    if (rc) return rc;
    if ((rc = jobs_idle(ctx))) return rc;
    if (nblocks == 0 && orig_len == 0) return w3_spec_validate(spec);
    if (!in || !block_lens || !out) return W3_E_INVALID;
    {   // any length: runs of blocks under the per-call limit, one after the other (W3_OPT_HOST_CHUNK_BLOCKS: the run length, for tests)
        const size_t run = ctx->host_chunk_blocks ? (size_t)ctx->host_chunk_blocks : host_call_cap_blocks(block_size);
        if (nblocks > run) {
            if ((uint64_t)(nblocks - 1) * block_size >= orig_len) { ctx->err = "more blocks than orig_len / block_size"; return W3_E_INVALID; }
            uint64_t coff = 0;
            for (size_t b0 = 0; b0 < nblocks; b0 += run) {
                const size_t b1 = std::min(nblocks, b0 + run);
                uint64_t clen = 0;
                for (size_t b = b0; b < b1; b++) clen += block_lens[b];
                if (coff + clen > in_len) { ctx->err = "block length table claims more compressed bytes than the buffer holds"; return W3_E_FORMAT; }
                const uint64_t o0 = (uint64_t)b0 * block_size, o1 = std::min<uint64_t>(orig_len, (uint64_t)b1 * block_size);
                const uint32_t keep = ctx->host_chunk_blocks;
                ctx->host_chunk_blocks = 0;   // (the pieces themselves are single calls)
                rc = w3_decode_blocks(ctx, spec, in + coff, (size_t)clen, block_lens + b0, b1 - b0, block_size, o1 - o0, out + o0);
                ctx->host_chunk_blocks = keep;
                if (rc) return rc;
                coff += clen;
            }
            return W3_OK;
        }
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    uint64_t total = 0;
    for (size_t b = 0; b < nblocks; b++) total += block_lens[b];
    if (total > in_len) { ctx->err = "block length table claims " + std::to_string(total) + " compressed bytes, the buffer holds " + std::to_string(in_len); return W3_E_FORMAT; }
    ENSURE(ctx, ctx->io_in, std::max<size_t>(total, 16));
    ENSURE(ctx, ctx->io_out, (size_t)orig_len);
    ENSURE(ctx, ctx->lens, nblocks * 4);
    HIPCHK(ctx, hipMemcpy(ctx->io_in.p, in, total, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->lens.p, block_lens, nblocks * 4, hipMemcpyHostToDevice));
    rc = w3_decode_blocks_device(ctx, spec, (const uint8_t *)ctx->io_in.p, (size_t)total, (const uint32_t *)ctx->lens.p, nblocks, block_size, orig_len,
                                 (uint8_t *)ctx->io_out.p, ctx->stream);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy(out, ctx->io_out.p, (size_t)orig_len, hipMemcpyDeviceToHost));
    return W3_OK;
}

// ---------------------------------------------------------------------------
// one process, several GPUs: contiguous block ranges, one host thread and context per device
// ---------------------------------------------------------------------------
extern "C" int w3_shard_range(size_t nblocks, int world, int rank, size_t *first_block, size_t *end_block) {
    if (world <= 0 || rank < 0 || rank >= world || !first_block || !end_block) return W3_E_INVALID;
    *first_block = (size_t)((unsigned __int128)nblocks * (unsigned)rank / (unsigned)world);
    *end_block = (size_t)((unsigned __int128)nblocks * (unsigned)(rank + 1) / (unsigned)world);
    return W3_OK;
}

extern "C" int w3_encode_blocks_sharded(w3_ctx *const *ctxs, int n_ctx, const w3_model_spec *spec, const uint8_t *in, size_t n, size_t block_size,
                                        uint8_t *out, size_t out_cap, size_t *out_len, uint32_t *block_lens) {
    if (!ctxs || n_ctx <= 0 || !out_len) return W3_E_INVALID;
    for (int r = 0; r < n_ctx; r++) {
        if (!ctxs[r]) return W3_E_INVALID;
        for (int q = 0; q < r; q++)
            if (ctxs[q] == ctxs[r]) { ctxs[0]->err = "the same context appears twice in ctxs[] (a context is not thread-safe)"; return W3_E_INVALID; }
    }
    *out_len = 0;
    int rc = check_args(ctxs[0], n, block_size, false);   // (the size limit of one call applies to each shard)
    if (rc) return rc;
    const size_t nb = (n + block_size - 1) / block_size;
    if (nb == 0) return w3_spec_validate(spec);
    if (!in || !block_lens) return W3_E_INVALID;
    struct Shard { size_t lo = 0, hi = 0, b0 = 0; std::vector<uint8_t> buf; size_t len = 0; int rc = W3_OK; };
    std::vector<Shard> sh(n_ctx);
    std::vector<std::thread> th;
    for (int r = 0; r < n_ctx; r++) {
        size_t b0, b1;
        (void)w3_shard_range(nb, n_ctx, r, &b0, &b1);
        sh[r].b0 = b0; sh[r].lo = std::min(b0 * block_size, n); sh[r].hi = std::min(b1 * block_size, n);
        if (sh[r].hi == sh[r].lo) continue;
        th.emplace_back([&, r]() {
            Shard &x = sh[r];
            const size_t m = x.hi - x.lo;
            size_t cap = 2 * m + 64 * ((m + block_size - 1) / block_size) + 64;   // realistic bound; grown to the reported need on W3_E_NOSPACE
            for (int attempt = 0; attempt < 2; attempt++) {
                x.buf.resize(cap);
                x.rc = w3_encode_blocks(ctxs[r], spec, in + x.lo, m, block_size, x.buf.data(), cap, &x.len, block_lens + x.b0);
                if (x.rc != W3_E_NOSPACE || x.len <= cap) break;
                cap = x.len;
            }
        });
    }
    for (auto &t : th) t.join();
    size_t total = 0;
    for (int r = 0; r < n_ctx; r++) {
        if (sh[r].rc) { ctxs[0]->err = "shard " + std::to_string(r) + ": " + (ctxs[r]->err.empty() ? w3_strerror(sh[r].rc) : ctxs[r]->err); return sh[r].rc; }
        total += sh[r].len;
    }
    *out_len = total;
    if (total > out_cap || !out) return W3_E_NOSPACE;
    size_t o = 0;
    for (int r = 0; r < n_ctx; r++) { if (sh[r].len) memcpy(out + o, sh[r].buf.data(), sh[r].len); o += sh[r].len; }
    return W3_OK;
}

// ---------------------------------------------------------------------------
// The same sharding with the data resident on the devices and the gather over xGMI (north star: "RCCL gather over xGMI to
// concatenate per-GPU compressed streams"; SURVEY section 8(e): ncclCommInitAll, sizes all-gather, grouped ncclSend / ncclRecv at
// the offsets of the exclusive scan — RCCL has no gatherv).  One process, one context and one host thread per device.
// ---------------------------------------------------------------------------
struct ShardComms {          // one communicator set per device list, created on first use and kept (ncclCommInitAll is slow)
    std::vector<int> devs;
    std::vector<w3rccl::comm_t> comms;
};
static std::mutex g_comm_mu;
static std::vector<ShardComms *> g_comms;

static ShardComms *shard_comms(const std::vector<int> &devs, std::string &err) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (ShardComms *c : g_comms)
        if (c->devs == devs) return c;
    w3rccl::Api *r = w3rccl::api();
    if (!r->error.empty()) { err = r->error; return nullptr; }
    ShardComms *c = new ShardComms();
    c->devs = devs; c->comms.assign(devs.size(), nullptr);
    const int rc = r->CommInitAll(c->comms.data(), (int)devs.size(), devs.data());
    if (rc != w3rccl::kSuccess) { err = std::string("ncclCommInitAll: ") + r->GetErrorString(rc); delete c; return nullptr; }
    g_comms.push_back(c);
    return c;
}

// The exchange step of a sharded encode: the ranks' totals (RCCL: an all-gather, checked against what the host knows), then the packed
// streams and length tables to the root at the exclusive scan of the totals / block counts.  src_*[r]: rank r's packed streams, length
// table and 8-byte total on ITS device; cm == nullptr: device copies (same device: D2D, other devices: peer copies) instead of RCCL.
// The transfers run on the contexts' own streams (idle while submitted calls are in flight) and have landed when this returns.
static int shard_gather(w3_ctx *const *ctxs, int n_ctx, int root, ShardComms *cm, void *const *src_out, void *const *src_lens, void *const *src_total,
                        const std::vector<size_t> &nbs, const uint64_t *totals, uint8_t *d_out, size_t out_cap, uint32_t *d_block_lens) {
    const bool use_rccl = cm != nullptr;
    w3_ctx *rt = ctxs[root];
    HIPCHK(rt, hipSetDevice(rt->device));
    // 2. the sizes: with RCCL an all-gather of every rank's total (the exchange step's first half, exercised even with one rank).
    // Nothing between ncclGroupStart and ncclGroupEnd may leave this function: an open group would swallow every later RCCL call of
    // this thread.  So whatever can fail for other reasons (allocations, memsets) is done first, and inside a group only RCCL's own
    // return codes are collected.
    if (use_rccl) {
        w3rccl::Api *rc_api = w3rccl::api();
        for (int r = 0; r < n_ctx; r++) {
            w3_ctx *c = ctxs[r];
            HIPCHK(c, hipSetDevice(c->device));
            if (!nbs[r]) HIPCHK(c, hipMemsetAsync(src_total[r], 0, 8, c->stream));   // (holds this rank's total already when the shard was not empty)
            ENSURE(c, c->misc, 8 * (size_t)n_ctx);
        }
        int gs = rc_api->GroupStart();
        const bool opened = gs == w3rccl::kSuccess;
        for (int r = 0; r < n_ctx && gs == w3rccl::kSuccess; r++) {
            (void)hipSetDevice(ctxs[r]->device);   // (a communicator knows its device; set for RCCL versions that look at the current one)
            gs = rc_api->AllGather(src_total[r], ctxs[r]->misc.p, 1, w3rccl::kUint64, cm->comms[r], ctxs[r]->stream);
        }
        const int ge = opened ? rc_api->GroupEnd() : w3rccl::kSuccess;
        if (gs != w3rccl::kSuccess || ge != w3rccl::kSuccess) { ctxs[0]->err = std::string("ncclAllGather: ") + rc_api->GetErrorString(gs != w3rccl::kSuccess ? gs : ge); return W3_E_HIP; }
        std::vector<uint64_t> seen(n_ctx);
        HIPCHK(rt, hipSetDevice(rt->device));
        HIPCHK(rt, hipStreamSynchronize(rt->stream));
        HIPCHK(rt, hipMemcpy(seen.data(), rt->misc.p, 8 * (size_t)n_ctx, hipMemcpyDeviceToHost));
        for (int r = 0; r < n_ctx; r++)
            if (seen[r] != totals[r]) { ctxs[0]->err = "sizes all-gather disagrees with the shards' totals (internal error)"; return W3_E_HIP; }
    }
    uint64_t sum = 0;
    for (int r = 0; r < n_ctx; r++) sum += totals[r];
    if (sum > out_cap) { ctxs[0]->err = "out_cap too small for the gathered streams"; return W3_E_NOSPACE; }

    // 3. the streams and length tables to the root, at the exclusive scan of the totals / block counts
    if (use_rccl) {
        w3rccl::Api *rc_api = w3rccl::api();
        HIPCHK(rt, hipSetDevice(rt->device));
        {   // the root's own shard: a device copy, outside the group
            uint64_t so = 0; size_t lo = 0;
            for (int r = 0; r < root; r++) { so += totals[r]; lo += nbs[r]; }
            if (totals[root]) HIPCHK(rt, hipMemcpyAsync(d_out + so, src_out[root], (size_t)totals[root], hipMemcpyDeviceToDevice, rt->stream));
            if (nbs[root]) HIPCHK(rt, hipMemcpyAsync(d_block_lens + lo, src_lens[root], nbs[root] * 4, hipMemcpyDeviceToDevice, rt->stream));
        }
        int gs = rc_api->GroupStart();
        const bool opened = gs == w3rccl::kSuccess;
        uint64_t so = 0; size_t lo = 0;
        for (int r = 0; r < n_ctx; r++) {
            w3_ctx *c = ctxs[r];
            if (r != root && nbs[r] && gs == w3rccl::kSuccess) {
                (void)hipSetDevice(c->device);
                // seven peers each have their own xGMI link to the root: the transfers of one group run concurrently
                gs = rc_api->Send(src_out[r], (size_t)totals[r], w3rccl::kUint8, root, cm->comms[r], c->stream);
                if (gs == w3rccl::kSuccess) gs = rc_api->Send(src_lens[r], nbs[r] * 4, w3rccl::kUint8, root, cm->comms[r], c->stream);
                (void)hipSetDevice(rt->device);
                if (gs == w3rccl::kSuccess) gs = rc_api->Recv(d_out + so, (size_t)totals[r], w3rccl::kUint8, r, cm->comms[root], rt->stream);
                if (gs == w3rccl::kSuccess) gs = rc_api->Recv(d_block_lens + lo, nbs[r] * 4, w3rccl::kUint8, r, cm->comms[root], rt->stream);
            }
            so += totals[r]; lo += nbs[r];
        }
        const int ge = opened ? rc_api->GroupEnd() : w3rccl::kSuccess;
        if (gs != w3rccl::kSuccess || ge != w3rccl::kSuccess) { ctxs[0]->err = std::string("RCCL gather: ") + rc_api->GetErrorString(gs != w3rccl::kSuccess ? gs : ge); return W3_E_HIP; }
        for (int r = 0; r < n_ctx; r++) {
            HIPCHK(ctxs[r], hipSetDevice(ctxs[r]->device));
            HIPCHK(ctxs[r], hipStreamSynchronize(ctxs[r]->stream));
        }
    } else {
        // device copies (same device: D2D; other devices: peer copies over xGMI / PCIe, no RCCL needed)
        uint64_t so = 0; size_t lo = 0;
        for (int r = 0; r < n_ctx; r++) {
            w3_ctx *c = ctxs[r];
            if (nbs[r]) {
                if (c->device == rt->device) {
                    HIPCHK(rt, hipMemcpyAsync(d_out + so, src_out[r], (size_t)totals[r], hipMemcpyDeviceToDevice, rt->stream));
                    HIPCHK(rt, hipMemcpyAsync(d_block_lens + lo, src_lens[r], nbs[r] * 4, hipMemcpyDeviceToDevice, rt->stream));
                } else {
                    HIPCHK(rt, hipMemcpyPeerAsync(d_out + so, rt->device, src_out[r], c->device, (size_t)totals[r], rt->stream));
                    HIPCHK(rt, hipMemcpyPeerAsync(d_block_lens + lo, rt->device, src_lens[r], c->device, nbs[r] * 4, rt->stream));
                }
            }
            so += totals[r]; lo += nbs[r];
        }
        HIPCHK(rt, hipStreamSynchronize(rt->stream));
    }
    return W3_OK;
}

extern "C" int w3_rccl_library(const char *path) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    if (w3rccl::resolved()) return W3_E_INVALID;   // (the library is resolved once per process, at the first gather or status call)
    w3rccl::library_override() = path ? path : "";
    return W3_OK;
}

extern "C" int w3_rccl_status(char *msg, size_t cap) {
    w3rccl::Api *r;
    { std::lock_guard<std::mutex> lk(g_comm_mu); r = w3rccl::api(); }
    if (msg && cap) { snprintf(msg, cap, "%s", r->error.empty() ? "RCCL resolved" : r->error.c_str()); }
    return r->error.empty() ? W3_OK : W3_E_HIP;
}

extern "C" int w3_encode_blocks_sharded_device(w3_ctx *const *ctxs, int n_ctx, const w3_model_spec *spec, const uint8_t *const *d_in, const size_t *n,
                                               size_t block_size, int root, uint8_t *d_out, size_t out_cap, uint32_t *d_block_lens,
                                               uint64_t *totals, int transport) {
    if (!ctxs || n_ctx <= 0 || n_ctx > 64 || !d_in || !n || !totals || root < 0 || root >= n_ctx) return W3_E_INVALID;
    if (transport < W3_GATHER_AUTO || transport > W3_GATHER_PEER_COPY) return W3_E_INVALID;
    std::vector<int> devs(n_ctx);
    bool distinct = true;
    size_t nb_total = 0;
    for (int r = 0; r < n_ctx; r++) {
        if (!ctxs[r]) return W3_E_INVALID;
        for (int q = 0; q < r; q++) {
            if (ctxs[q] == ctxs[r]) { ctxs[0]->err = "the same context appears twice in ctxs[] (a context is not thread-safe)"; return W3_E_INVALID; }
            distinct &= ctxs[q]->device != ctxs[r]->device;
        }
        devs[r] = ctxs[r]->device;
        totals[r] = 0;
        int rc = check_args(ctxs[r], n[r], block_size);
        if (rc) return rc;
        if (n[r] && !d_in[r]) return W3_E_INVALID;
        if (r + 1 < n_ctx && n[r] % block_size) { ctxs[0]->err = "every shard but the last must be a whole number of blocks (w3_shard_range)"; return W3_E_INVALID; }
        if ((rc = jobs_idle(ctxs[r]))) return rc;
        nb_total += (n[r] + block_size - 1) / block_size;
    }
    if (nb_total == 0) return w3_spec_validate(spec);
    if (!d_out || !d_block_lens) return W3_E_INVALID;
    // RCCL needs one device per rank (ncclCommInitAll refuses a device twice): contexts that share a device — how the path is
    // rehearsed on a 1-GPU box — gather with device copies instead
    const bool use_rccl = transport == W3_GATHER_RCCL || (transport == W3_GATHER_AUTO && distinct && n_ctx > 1);
    if (use_rccl && !distinct) { ctxs[0]->err = "W3_GATHER_RCCL needs one device per context"; return W3_E_INVALID; }
    ShardComms *cm = nullptr;
    if (use_rccl && !(cm = shard_comms(devs, ctxs[0]->err))) return W3_E_HIP;

    // 1. every shard on its own device and host thread, into its context's staging buffers (io_out, lens, total)
    std::vector<int> rcs(n_ctx, W3_OK);
    std::vector<size_t> nbs(n_ctx, 0);
    {
        std::vector<std::thread> th;
        for (int r = 0; r < n_ctx; r++) {
            nbs[r] = (n[r] + block_size - 1) / block_size;
            if (!nbs[r]) continue;
            th.emplace_back([&, r]() {
                w3_ctx *c = ctxs[r];
                auto body = [&]() -> int {
                    HIPCHK(c, hipSetDevice(c->device));
                    size_t cap = n[r] + n[r] / 4 + 64 * nbs[r] + 1024;   // realistic bound; grown to the reported need on W3_E_NOSPACE
                    for (int attempt = 0; attempt < 2; attempt++) {
                        ENSURE(c, c->io_out, cap);
                        ENSURE(c, c->lens, nbs[r] * 4);
                        ENSURE(c, c->total, 8);
                        int rc = encode_core(c, jobref(c, 0), spec, d_in[r], n[r], block_size, (uint8_t *)c->io_out.p, cap, (uint32_t *)c->lens.p, (uint64_t *)c->total.p, nullptr);
                        uint64_t t = 0;
                        if (rc == W3_OK || rc == W3_E_NOSPACE) HIPCHK(c, hipMemcpy(&t, c->total.p, 8, hipMemcpyDeviceToHost));
                        totals[r] = t;
                        if (rc != W3_E_NOSPACE || t <= cap) return rc;
                        cap = (size_t)t;
                    }
                    return W3_E_NOSPACE;
                };
                rcs[r] = body();
            });
        }
        for (auto &t : th) t.join();
    }
    for (int r = 0; r < n_ctx; r++)
        if (rcs[r]) { if (r) ctxs[0]->err = "shard " + std::to_string(r) + ": " + (ctxs[r]->err.empty() ? w3_strerror(rcs[r]) : ctxs[r]->err); return rcs[r]; }

    std::vector<void *> so(n_ctx), sl(n_ctx), st(n_ctx);
    for (int r = 0; r < n_ctx; r++) {
        w3_ctx *c = ctxs[r];
        HIPCHK(c, hipSetDevice(c->device));
        ENSURE(c, c->total, 8);
        so[r] = c->io_out.p; sl[r] = c->lens.p; st[r] = c->total.p;
    }
    return shard_gather(ctxs, n_ctx, root, cm, so.data(), sl.data(), st.data(), nbs, totals, d_out, out_cap, d_block_lens);
}

// ---------------------------------------------------------------------------
// The sharded encode as a STREAM of steps (ABI v8): every context keeps w3_encode_max_in_flight calls in flight on its device
// (w3_encode_submit), and a step's packed streams are gathered on the root when the step is waited for — the exchange step of step k
// runs on the contexts' own streams while the devices are already coding step k+1.  What bench.py --gpus N does through
// torch.distributed (one process per GPU), for a host that is ONE process.
// ---------------------------------------------------------------------------
static int sharded_args(w3_ctx *const *ctxs, int n_ctx, std::vector<int> &devs, bool &distinct) {
    if (!ctxs || n_ctx <= 0 || n_ctx > 64) return W3_E_INVALID;
    devs.assign(n_ctx, 0);
    distinct = true;
    for (int r = 0; r < n_ctx; r++) {
        if (!ctxs[r]) return W3_E_INVALID;
        for (int q = 0; q < r; q++) {
            if (ctxs[q] == ctxs[r]) { ctxs[0]->err = "the same context appears twice in ctxs[] (a context is not thread-safe)"; return W3_E_INVALID; }
            distinct &= ctxs[q]->device != ctxs[r]->device;
        }
        devs[r] = ctxs[r]->device;
    }
    return W3_OK;
}

extern "C" int w3_encode_sharded_max_in_flight(const w3_model_spec *spec, const size_t *n, int n_ctx, size_t block_size) {
    if (!n || n_ctx <= 0 || !block_size) return 0;
    int depth = W3_MAX_JOBS;
    for (int r = 0; r < n_ctx; r++)
        if (n[r]) depth = std::min(depth, w3_encode_max_in_flight(spec, n[r], block_size));
    return depth;
}

extern "C" int w3_encode_sharded_submit(w3_ctx *const *ctxs, int n_ctx, const w3_model_spec *spec, const uint8_t *const *d_in, const size_t *n,
                                        size_t block_size, int *sjob) {
    if (!sjob || !d_in || !n) return W3_E_INVALID;
    *sjob = -1;
    std::vector<int> devs;
    bool distinct;
    int rc = sharded_args(ctxs, n_ctx, devs, distinct);
    if (rc) return rc;
    ParsedSpec ps;
    if ((rc = parse_spec(spec, ps))) { ctxs[0]->err = "malformed model spec"; return rc; }
    size_t nb_total = 0;
    for (int r = 0; r < n_ctx; r++) {
        if ((rc = check_args(ctxs[r], n[r], block_size))) return rc;
        if (n[r] && !d_in[r]) return W3_E_INVALID;
        if (r + 1 < n_ctx && n[r] % block_size) { ctxs[0]->err = "every shard but the last must be a whole number of blocks (w3_shard_range)"; return W3_E_INVALID; }
        const size_t nb = (n[r] + block_size - 1) / block_size;
        nb_total += nb;
        // (a shard that w3_encode_submit runs synchronously — a ragged tail below 8 bytes, a lane-per-block spec — is coded inside this
        // call, one context after the other: correct, but such specs are better served by the one-shot form, which codes the shards from
        // one host thread each)
        for (const auto &h : ctxs[r]->hj)
            if (h.state != 0) { ctxs[0]->err = "host-buffer jobs are in flight on a context"; return W3_E_INVALID; }
    }
    if (nb_total == 0) { ctxs[0]->err = "nothing to encode"; return W3_E_INVALID; }
    // the same slot on every context; as many steps in flight as the smallest w3_encode_max_in_flight among the shards
    int slot = -1, busy = 0;
    for (int k = 0; k < W3_MAX_JOBS; k++) {
        bool free_everywhere = true, used = false;
        for (int r = 0; r < n_ctx; r++) { free_everywhere &= ctxs[r]->ss[k].state == 0; used |= ctxs[r]->ss[k].state != 0; }
        busy += used;
        if (free_everywhere && slot < 0) slot = k;
    }
    const int depth = w3_encode_sharded_max_in_flight(spec, n, n_ctx, block_size);
    if (slot < 0 || busy >= depth) {
        ctxs[0]->err = std::to_string(busy) + " sharded steps are in flight already (at most " + std::to_string(depth) + " for shards of this size): w3_encode_sharded_wait the oldest one first";
        return W3_E_INVALID;
    }
    for (int r = 0; r < n_ctx; r++) {
        w3_ctx *c = ctxs[r];
        w3_ctx::ShardSlot &x = c->ss[slot];
        auto body = [&]() -> int {
            HIPCHK(c, hipSetDevice(c->device));
            x.nb = (n[r] + block_size - 1) / block_size; x.n = n[r]; x.d_in = d_in[r]; x.block_size = block_size; x.djob = -1;
            x.spec = *spec;
            if (ps.n_huff) { memcpy(x.huff_copy, ps.huff, sizeof(w3_huff_table) * ps.n_huff); x.spec.huff = x.huff_copy; }
            ENSURE(c, x.total, 8);
            if (!x.nb) { HIPCHK(c, hipMemsetAsync(x.total.p, 0, 8, c->stream)); return W3_OK; }
            x.cap = n[r] + n[r] / 4 + 64 * x.nb + 1024;   // realistic bound; a shard beyond it is redone with the room it asks for (the wait)
            ENSURE(c, x.out, x.cap);
            ENSURE(c, x.lens, x.nb * 4);
            return w3_encode_submit(c, &x.spec, d_in[r], n[r], block_size, (uint8_t *)x.out.p, x.cap, (uint32_t *)x.lens.p, (uint64_t *)x.total.p, nullptr, &x.djob);
        };
        rc = body();
        if (rc) {   // leave nothing in flight behind an error: the shards submitted so far are completed and dropped
            if (r) ctxs[0]->err = "shard " + std::to_string(r) + ": " + (c->err.empty() ? w3_strerror(rc) : c->err);
            for (int q = 0; q < r; q++) {
                w3_ctx::ShardSlot &y = ctxs[q]->ss[slot];
                if (y.djob >= 0) (void)w3_encode_wait(ctxs[q], y.djob);
                y.state = 0; y.djob = -1;
            }
            return rc;
        }
        x.state = 1;
    }
    *sjob = slot;
    return W3_OK;
}

extern "C" int w3_encode_sharded_wait(w3_ctx *const *ctxs, int n_ctx, int sjob, int root, uint8_t *d_out, size_t out_cap, uint32_t *d_block_lens,
                                      uint64_t *totals, int transport) {
    if (!totals || sjob < 0 || sjob >= W3_MAX_JOBS || root < 0 || root >= n_ctx) return W3_E_INVALID;
    if (transport < W3_GATHER_AUTO || transport > W3_GATHER_PEER_COPY) return W3_E_INVALID;
    std::vector<int> devs;
    bool distinct;
    int rc = sharded_args(ctxs, n_ctx, devs, distinct);
    if (rc) return rc;
    for (int r = 0; r < n_ctx; r++)
        if (ctxs[r]->ss[sjob].state != 1) { ctxs[0]->err = "no such sharded step in flight"; return W3_E_INVALID; }
    if (!d_out || !d_block_lens) return W3_E_INVALID;
    const bool use_rccl = transport == W3_GATHER_RCCL || (transport == W3_GATHER_AUTO && distinct && n_ctx > 1);
    if (use_rccl && !distinct) { ctxs[0]->err = "W3_GATHER_RCCL needs one device per context"; return W3_E_INVALID; }
    // 1. the shards' encodes (the devices run them concurrently; the waits only read pinned status words)
    std::vector<size_t> nbs(n_ctx, 0);
    int first_rc = W3_OK;
    for (int r = 0; r < n_ctx; r++) {
        w3_ctx *c = ctxs[r];
        w3_ctx::ShardSlot &x = c->ss[sjob];
        nbs[r] = x.nb;
        totals[r] = 0;
        auto body = [&]() -> int {
            if (x.djob < 0) return W3_OK;
            int rc1 = w3_encode_wait(c, x.djob);
            uint64_t t = 0;
            const JobState &ds = c->js[x.djob];
            if (rc1 == W3_OK || rc1 == W3_E_NOSPACE) {
                if (ds.total_valid) t = ds.total_out;
                else { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipMemcpy(&t, x.total.p, 8, hipMemcpyDeviceToHost)); }
            }
            if (rc1 == W3_E_NOSPACE && t > x.cap) {   // beyond the realistic bound: once more, alone, with the room it asked for
                HIPCHK(c, hipSetDevice(c->device));
                HIPCHK(c, hipDeviceSynchronize());
                ENSURE(c, x.out, (size_t)t);
                x.cap = (size_t)t;
                rc1 = encode_core(c, jobref(c, x.djob), &x.spec, x.d_in, x.n, x.block_size, (uint8_t *)x.out.p, x.cap, (uint32_t *)x.lens.p, (uint64_t *)x.total.p, c->stream);
                if (rc1 == W3_OK) HIPCHK(c, hipMemcpy(&t, x.total.p, 8, hipMemcpyDeviceToHost));
            }
            totals[r] = t;
            return rc1;
        };
        const int rc1 = body();
        if (rc1 && first_rc == W3_OK) { first_rc = rc1; if (r) ctxs[0]->err = "shard " + std::to_string(r) + ": " + (c->err.empty() ? w3_strerror(rc1) : c->err); }
    }
    // 2. the exchange step
    int grc = first_rc;
    if (grc == W3_OK) {
        ShardComms *cm = nullptr;
        if (use_rccl && !(cm = shard_comms(devs, ctxs[0]->err))) grc = W3_E_HIP;
        if (grc == W3_OK) {
            std::vector<void *> so(n_ctx), sl(n_ctx), st(n_ctx);
            for (int r = 0; r < n_ctx; r++) { w3_ctx::ShardSlot &x = ctxs[r]->ss[sjob]; so[r] = x.out.p; sl[r] = x.lens.p; st[r] = x.total.p; }
            grc = shard_gather(ctxs, n_ctx, root, cm, so.data(), sl.data(), st.data(), nbs, totals, d_out, out_cap, d_block_lens);
        }
    }
    for (int r = 0; r < n_ctx; r++) { ctxs[r]->ss[sjob].state = 0; ctxs[r]->ss[sjob].djob = -1; }
    return grc;
}

// ---------------------------------------------------------------------------
// reference container: b"w30i" + u64 BE len + one stream  (main.rs:14-15, 89-144)
// ---------------------------------------------------------------------------
extern "C" int w3_compress_stream(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *in, size_t n, uint8_t *out, size_t out_cap,
                                  size_t *out_len) {
    if (!ctx || !out_len) return W3_E_INVALID;
    *out_len = 0;
    // One stream is one serial chain = ONE GPU lane (~260 ns per bit-step: 2^28 bytes take ~10 minutes); larger inputs belong in the block
    // container, which is what the device is for — the reference's format has no blocks to code in parallel.
    if (n > (1u << 28)) { ctx->err = "the w30i single-stream container is limited to 2^28 bytes on the device (one serial chain = one GPU lane): use the block container (w3_encode_blocks; tools/w3cli without W3_CONTAINER=w30i) for larger inputs"; return W3_E_UNSUPPORTED; }
    int rc = w3_spec_validate(spec);
    if (rc) return rc;
    uint8_t hdr[12] = {'w', '3', '0', 'i'};
    for (int i = 0; i < 8; i++) hdr[4 + i] = (uint8_t)((uint64_t)n >> (8 * (7 - i)));
    size_t body = 0;
    if (n == 0) {
        // empty file: the coder still flushes x2 = 0xFFFFFFFF -> one 0xFF byte (io.rs:91-100)
        *out_len = 13;
        if (out_cap < 13 || !out) return W3_E_NOSPACE;
        memcpy(out, hdr, 12);
        out[12] = 0xFF;
        return W3_OK;
    }
    uint32_t blen = 0;
    rc = w3_encode_blocks(ctx, spec, in, n, n, out_cap > 12 && out ? out + 12 : nullptr, out_cap > 12 ? out_cap - 12 : 0, &body, &blen);
    *out_len = body + 12;
    if (rc) return rc;
    memcpy(out, hdr, 12);
    return W3_OK;
}

extern "C" int w3_decompress_stream(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *in, size_t in_len, uint8_t *out,
                                    size_t out_cap, size_t *out_len) {
    if (!ctx || !in || !out_len) return W3_E_INVALID;
    *out_len = 0;
    if (in_len < 12) { ctx->err = "truncated header"; return W3_E_FORMAT; }
    if (memcmp(in, "w30i", 4) != 0) { ctx->err = "Magic numbers don't match up"; return W3_E_FORMAT; }
    uint64_t len = 0;
    for (int i = 0; i < 8; i++) len = (len << 8) | in[4 + i];
    *out_len = (size_t)len;
    if (len == 0) return w3_spec_validate(spec);
    if (len > (1u << 28)) { ctx->err = "the w30i single-stream container is limited to 2^28 bytes on the device (one serial chain = one GPU lane); larger inputs use the block container (w3_decode_blocks)"; return W3_E_UNSUPPORTED; }
    if (len > out_cap || !out) return W3_E_NOSPACE;
    uint32_t blen = (uint32_t)(in_len - 12);
    uint8_t zero = 0;
    const uint8_t *body = blen ? in + 12 : &zero;  // ACReader pads with zeros past EOF (io.rs:23-26)
    return w3_decode_blocks(ctx, spec, body, blen, &blen, 1, (size_t)len, len, out);
}

// ---------------------------------------------------------------------------
// StationaryModel::new (models/ac_hash/stationary.rs:14-34): constructor-time
// table prep on the host (8 Counters, one per bit position, index 0 = MSB).
// ---------------------------------------------------------------------------
extern "C" int w3_stationary_table(const uint8_t *buf, size_t n, uint16_t table[8]) {
    if ((!buf && n) || !table) return W3_E_INVALID;
    uint32_t c0[8] = {0}, c1[8] = {0};
    for (size_t k = 0; k < n; k++) {
        for (int i = 0; i < 8; i++) {
            uint32_t bit = (buf[k] >> (7 - i)) & 1u;
            uint32_t &c = bit ? c1[i] : c0[i];
            if (++c == 0xFFFFu) {  // Counter::update halves BOTH counts (counter.rs:22-25)
                c0[i] = (c0[i] >> 1) + (c0[i] & 1u);
                c1[i] = (c1[i] >> 1) + (c1[i] & 1u);
            }
        }
    }
    for (int i = 0; i < 8; i++) {
        uint64_t p = (1ull << 17) * ((uint64_t)c1[i] + 1) / ((uint64_t)c0[i] + c1[i] + 2);
        table[i] = (uint16_t)((p >> 1) + (p & 1));
    }
    return W3_OK;
}

// ---------------------------------------------------------------------------
// OrderN(bits, align) parameter sweep, configurations x blocks in one launch (w3_sweep.h; bin/ordern/main.rs:9-80)
// ---------------------------------------------------------------------------
extern "C" int w3_sweep_ordern_device(w3_ctx *ctx, const uint8_t *d_in, size_t n, size_t block_size, const uint8_t *bits, const uint8_t *aligns,
                                      size_t ncfg, uint32_t *block_bits) {
    int rc = check_args(ctx, n, block_size);
    if (rc) return rc;
    if ((rc = jobs_idle(ctx))) return rc;
    const uint32_t nb = (uint32_t)((n + block_size - 1) / block_size);
    if (nb == 0 || ncfg == 0) return W3_OK;
    if (!d_in || !bits || !aligns || !block_bits || ncfg > 4096) return W3_E_INVALID;
    for (size_t c = 0; c < ncfg; c++)   // OrderN::new allocates 1 << bits counters; masks are u32 (ordern.rs:35-43)
        if (bits[c] < 1 || bits[c] > 32 || aligns[c] > 7 || aligns[c] > bits[c] || (int)bits[c] - (int)aligns[c] > 31) return W3_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t steps = (uint64_t)block_size * 8;
    const uint64_t hash_slots = std::max<uint64_t>(1024, next_pow2(2 * steps)), hash_bytes = hash_slots * 8;
    std::vector<SweepCfg> cfg(ncfg);
    size_t free_b = 0, total_b = 0;
    HIPCHK(ctx, hipMemGetInfo(&free_b, &total_b));
    const uint64_t budget = std::min<uint64_t>((uint64_t)(free_b + ctx->tables.cap) * 3 / 4, 200ull << 30);
    ENSURE(ctx, ctx->sweep, ncfg * sizeof(SweepCfg) + (size_t)ncfg * nb * 4);
    SweepCfg *d_cfg = (SweepCfg *)ctx->sweep.p;
    uint32_t *d_bits = (uint32_t *)((uint8_t *)ctx->sweep.p + ncfg * sizeof(SweepCfg));
    SweepArgs a;
    memset(&a, 0, sizeof a);
    a.in = d_in; a.n = n; a.block_size = (uint32_t)block_size; a.nblocks = nb; a.waves_per_cfg = (nb + 63) / 64; a.ncfg = (uint32_t)ncfg;
    a.cfg = d_cfg; a.out_bits = d_bits;
    size_t c0 = 0;
    while (c0 < ncfg) {   // as many configurations per launch as their tables fit the budget
        uint64_t used = 0;
        size_t c1 = c0;
        for (; c1 < ncfg; c1++) {
            const uint64_t direct = 4ull << bits[c1];
            const bool hashed = direct > hash_bytes;
            const uint64_t stride = hashed ? hash_bytes : std::max<uint64_t>(direct, 16);
            if (c1 > c0 && used + stride * nb > budget) break;
            SweepCfg &cf = cfg[c1];
            cf.bits = bits[c1]; cf.align = aligns[c1]; cf.use_hash = hashed; cf.pad = 0;
            cf.hash_mask = (uint32_t)(hash_slots - 1); cf.hist_mask = (uint32_t)((1ull << (bits[c1] - aligns[c1])) - 1ull);
            cf.base = used; cf.stride = stride;
            used += stride * nb;
        }
        if (used > (uint64_t)(free_b + ctx->tables.cap)) { ctx->err = "sweep tables of one configuration do not fit the device"; return W3_E_NOMEM; }
        ENSURE(ctx, ctx->tables, (size_t)used);
        HIPCHK(ctx, hipMemsetAsync(ctx->tables.p, 0, (size_t)used, s));
        HIPCHK(ctx, hipMemcpyAsync(d_cfg + c0, cfg.data() + c0, (c1 - c0) * sizeof(SweepCfg), hipMemcpyHostToDevice, s));
        a.tables = (uint8_t *)ctx->tables.p; a.first_cfg = (uint32_t)c0;
        hipLaunchKernelGGL(k_sweep_ordern, dim3((unsigned)((c1 - c0) * a.waves_per_cfg)), dim3(64), 0, s, a);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(s));   // (cfg is host memory reused by the next batch; the tables are re-zeroed)
        c0 = c1;
    }
    HIPCHK(ctx, hipMemcpy(block_bits, d_bits, (size_t)ncfg * nb * 4, hipMemcpyDeviceToHost));
    return W3_OK;
}

extern "C" int w3_sweep_ordern(w3_ctx *ctx, const uint8_t *in, size_t n, size_t block_size, const uint8_t *bits, const uint8_t *aligns,
                               size_t ncfg, uint32_t *block_bits) {
    int rc = check_args(ctx, n, block_size);
    if (rc) return rc;
    if ((rc = jobs_idle(ctx))) return rc;
    if (n == 0 || ncfg == 0) return W3_OK;
    if (!in) return W3_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ENSURE(ctx, ctx->io_in, n);
    HIPCHK(ctx, hipMemcpy(ctx->io_in.p, in, n, hipMemcpyHostToDevice));
    return w3_sweep_ordern_device(ctx, (const uint8_t *)ctx->io_in.p, n, block_size, bits, aligns, ncfg, block_bits);
}

// ---------------------------------------------------------------------------
// Context statistics export (README.md:9 "output stats from contexts for use by external neural nets"): the Counter table of a
// Counter-table model after it has seen the whole input as ONE stream — what `stats` of models/ordern.rs:5 holds when the
// reference's compress() returns.  One serial chain (one lane, as w3_compress_stream).
// ---------------------------------------------------------------------------
extern "C" int w3_export_counters(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *in, size_t n, uint32_t *counters) {
    if (!ctx || !counters) return W3_E_INVALID;
    ParsedSpec ps;
    int rc = jobs_idle(ctx);
    if (rc) return rc;
    if ((rc = parse_spec(spec, ps))) return rc;
    if (ps.n_leaves != 1 || ps.n_apm || ps.has_slot || ps.leaf[0].frozen) { ctx->err = "w3_export_counters takes ONE adaptive Counter-table leaf"; return W3_E_UNSUPPORTED; }
    const w3_node &nd = ps.leaf[0];
    if (nd.bits > 28) { ctx->err = "tables above 2^28 counters are not exported"; return W3_E_UNSUPPORTED; }
    if (n > (1u << 28)) { ctx->err = "single-stream export limited to 2^28 bytes"; return W3_E_UNSUPPORTED; }
    const size_t entries = (size_t)1 << nd.bits;
    if (n == 0) { memset(counters, 0, entries * 4); return W3_OK; }
    if (!in) return W3_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if ((rc = stage_huff(ctx, s, ps))) return rc;
    GenericArgs ga;
    memset(&ga, 0, sizeof ga);
    (void)layout_generic(ps, n, ga);
    ga.leaf[0].use_hash = 0; ga.leaf[0].tbl_off = 0;            // direct-indexed, like the reference's Vec<Counter>
    if ((rc = prepare_achash_luts(ctx, s, ga))) return rc;
    const uint32_t cap = default_stripe_cap(n);
    ENSURE(ctx, ctx->tables, entries * 4);
    ENSURE(ctx, ctx->io_in, n);
    ENSURE(ctx, ctx->stripes, cap);
    ENSURE(ctx, ctx->lens, 4);
    ENSURE(ctx, ctx->flag, 16);
    HIPCHK(ctx, hipMemcpyAsync(ctx->io_in.p, in, n, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemsetAsync(ctx->tables.p, 0, entries * 4, s));
    ga.n = n; ga.block_size = (uint32_t)n; ga.first_block = 0; ga.n_lanes = 1;
    ga.huff = ctx->tp.huff; ga.n_huff = (int)ps.n_huff;
    ga.tables = (uint8_t *)ctx->tables.p; ga.lane_stride = entries * 4;
    ga.in = (const uint8_t *)ctx->io_in.p; ga.stripes = (uint8_t *)ctx->stripes.p; ga.stripe_cap = cap;
    ga.out_len = (uint32_t *)ctx->lens.p; ga.overflow = (uint32_t *)ctx->flag.p; ga.out_bits = nullptr;
    hipLaunchKernelGGL((k_generic_nl<false, 1>), dim3(1), dim3(64), 0, s, ga);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(counters, ctx->tables.p, entries * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return W3_OK;
}

// ---------------------------------------------------------------------------
// HuffHistory::new (history/huff_history.rs:17-55): constructor-time table prep on the host (w3_huff.h)
// ---------------------------------------------------------------------------
extern "C" int w3_huff_tables(const uint8_t *buf, size_t n, uint8_t huff_size, uint8_t rem_huff_size, w3_huff_table *out) {
    if ((!buf && n) || !out) return W3_E_INVALID;
    if (huff_size > 16 || rem_huff_size > 16) return W3_E_INVALID;   // codes are u16 (package_merge.rs:87: Vec<(u16, u8)>)
    return w3huff::build(buf, n, huff_size, rem_huff_size, out) ? W3_OK : W3_E_INVALID;
}

// ---------------------------------------------------------------------------
// Model::predict for every step (two-phase predict kernels only)
// ---------------------------------------------------------------------------
extern "C" int w3_predict_blocks(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *in, size_t n, size_t block_size, uint16_t *p_out) {
    int rc = check_args(ctx, n, block_size);
    if (rc) return rc;
    if ((rc = jobs_idle(ctx))) return rc;   // (the predict phase runs on job 0's workspace)
    ParsedSpec ps;
    if ((rc = parse_spec(spec, ps))) return rc;
    if (n == 0) return W3_OK;
    if (!in || !p_out) return W3_E_INVALID;
    if (!twophase_supported(ps, block_size, n)) { ctx->err = "spec not covered by the two-phase predict kernels"; return W3_E_UNSUPPORTED; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (ps.is_cm()) {
        CmArgs lut;
        if ((rc = cm_luts(ctx, s, lut))) return rc;
        ctx->tp.stretch = lut.stretch; ctx->tp.squash = lut.squash; ctx->tp.st = lut.st;
    }
    const uint32_t nb = (uint32_t)((n + block_size - 1) / block_size);
    if ((rc = stage_huff(ctx, s, ps))) return rc;
    ENSURE(ctx, ctx->io_in, n);
    HIPCHK(ctx, hipMemcpyAsync(ctx->io_in.p, in, n, hipMemcpyHostToDevice, s));
    const uint16_t *d_p = nullptr;
    { JobRef J0 = jobref(ctx, 0); if ((rc = attach_aux_streams(ctx, J0, 0))) return rc; }
    rc = twophase_predict(ctx->tp, s, ps, (const uint8_t *)ctx->io_in.p, n, block_size, nb, ps.n_apm == 0, &d_p, nullptr, &ctx->timing, ctx->err);
    if (rc) return rc;
    if ((rc = twophase_apm(ctx->tp, s, ps, (const uint8_t *)ctx->io_in.p, n, block_size, nb, nullptr, &ctx->timing, ctx->err))) return rc;
    d_p = (const uint16_t *)ctx->tp.P;
    HIPCHK(ctx, hipStreamSynchronize(s));
    HIPCHK(ctx, hipMemcpy(p_out, d_p, n * 16, hipMemcpyDeviceToHost));
    return W3_OK;
}

// ---------------------------------------------------------------------------
// read-only tables (host-side known-answer surface)
// ---------------------------------------------------------------------------
extern "C" int w3_state_table(uint16_t *out) {
    if (!out) return W3_E_INVALID;
    std::vector<StEntry> t(kStSize);
    build_state_table(t.data());
    for (int i = 0; i < kStSize; i++) { out[3 * i] = t[i].prob; out[3 * i + 1] = t[i].next0; out[3 * i + 2] = t[i].next1; }
    return W3_OK;
}

extern "C" int w3_stretch_squash(int16_t *stretch, uint16_t *squash) {
    if (!stretch || !squash) return W3_E_INVALID;
    build_stretch_squash(stretch, squash);
    return W3_OK;
}

// ---------------------------------------------------------------------------
// device self-test
// ---------------------------------------------------------------------------
extern "C" int w3_selftest_counter_p(w3_ctx *ctx, uint64_t *mismatches) {
    if (!ctx || !mismatches) return W3_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ENSURE(ctx, ctx->total, 8);
    HIPCHK(ctx, hipMemsetAsync(ctx->total.p, 0, 8, ctx->stream));
    for (uint32_t lo = 0; lo < 65536; lo += 4096) {
        hipLaunchKernelGGL(k_selftest_counter_p, dim3(256 * 16), dim3(256), 0, ctx->stream, (unsigned long long *)ctx->total.p, lo, lo + 4096);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipMemcpyAsync(mismatches, ctx->total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return W3_OK;
}

extern "C" int w3_debug_get_stamps(w3_ctx *ctx, uint64_t out[8]) {
    if (!ctx || !out) return W3_E_INVALID;
    memset(out, 0, 64);
    if (!ctx->tp.dbg) return W3_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemcpy(out, ctx->tp.dbg, 64, hipMemcpyDeviceToHost));
    return W3_OK;
}
