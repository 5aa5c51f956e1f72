/*
 * w3hip.h — C ABI of the MI355X-native weath3rb0i hot path (libw3hip.so).
 *
 * The reference (mitiko/weath3rb0i) is a CPU-only Rust crate with no FFI or
 * plugin interface; its seams are Rust traits and two private functions in
 * src/main.rs.  This header is the drop-in boundary a Rust shim would bind
 * (see INTEGRATION.md for the `extern "C"` block).  Each entry point cites
 * the reference interface it replaces (paths under /root/reference/src).
 *
 * Conventions: plain pointers and sizes; caller owns every buffer; every call
 * returns 0 (W3_OK) or a negative W3_E_* code — nothing panics or throws
 * across the ABI (the reference aborts on panic, Cargo.toml:24).  A w3_ctx is
 * bound to one GPU and is NOT thread-safe (the reference is single-threaded,
 * Cargo.toml:14-15).  There is no CPU fallback: without a HIP device
 * w3_ctx_create fails with W3_E_HIP.
 */
#ifndef W3HIP_H
#define W3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define W3_ABI_VERSION 8

/* ---- error codes --------------------------------------------------------- */
enum {
    W3_OK            =  0,
    W3_E_INVALID     = -1,  /* bad argument / malformed model spec                    */
    W3_E_NOSPACE     = -2,  /* out_cap too small; *out_len holds the size needed      */
    W3_E_HIP         = -3,  /* HIP runtime error (w3_last_error has the text)         */
    W3_E_UNSUPPORTED = -4,  /* valid spec the device path does not implement          */
    W3_E_NOMEM       = -5,  /* device workspace does not fit                          */
    W3_E_FORMAT      = -6   /* bad container magic (main.rs:123-124 assert_eq!) or a block length table that
                               claims more compressed bytes than the input buffer holds */
};

/* ---- model spec ----------------------------------------------------------
 * Mirrors the compile-time composition done in init_model() (main.rs:146-152)
 * as a caller-owned, read-only POD: the model tree in POSTFIX order.
 *   leaf  W3_NODE_ORDERN : OrderN::new(bits, align)              models/ordern.rs:14-23
 *                          history = W3_HIST_RAW / W3_HIST_AC / W3_HIST_HUFF makes it
 *                          OrderNEntropy::new(bits, align, hist) models/ordern_entropy.rs:15-24
 *                          (RawHistory history/raw_history.rs; ACHistory::new(max_bits,
 *                          StationaryModel::from_table(table)) history/ac_history.rs:16-19;
 *                          HuffHistory history/huff_history.rs:9-76 with the two code tables
 *                          spec->huff[node.reserved], see w3_huff_table below)
 *                          frozen=1 wraps it in FrozenModel      models/frozen.rs:7-11
 *                          Order0 == (11,3), Order1 == (19,3)    models/order0.rs, order1.rs
 *                          (bijective re-indexing, bin/cmp/main.rs:14-24)
 *   node  W3_NODE_BEST_OF_TWO : BestOfTwoModel::new(a, b)        models/mod.rs:42-75
 *                          pops the two preceding subtrees (a pushed first).
 *
 * BUILD-DEFINED nodes (SURVEY §8 A19 ii-v: the reference ships the primitives but no model
 * that uses them; README.md:6-10 lists "12-bit state table" and "APM mixers" as goals):
 *   leaf  W3_NODE_SLOT_STATE : state-table CM leaf per docs/hashslots.md — context = the
 *                          previous `bits` (= order, 0..7) bytes, hashed once per nibble into a
 *                          HashMap of 2^log_cells Cells (hashmap.rs:1-71); each Slot holds the 15
 *                          12-bit NaiveStateTable states of one nibble (hashmap.rs:73-129,
 *                          state_table/naive.rs); P(1) = StateTable::p(state), update =
 *                          StateTable::next.  Tag miss = the reference's TODO (hashmap.rs:64-68):
 *                          least-observed slot is overwritten.
 *   node  W3_NODE_APM       : adaptive probability map over stretch(p) ("APM mixers"): pops one
 *                          subtree.  align = context kind (W3_APM_ORDER0: partial byte, 256 rows;
 *                          W3_APM_ORDER1: partial byte | previous byte << 8, 65536 rows),
 *                          max_bits = adaptation rate (1..15).  Only as a chain at the root.
 */
enum { W3_NODE_ORDERN = 1, W3_NODE_BEST_OF_TWO = 2, W3_NODE_SLOT_STATE = 3, W3_NODE_APM = 4 };
enum { W3_HIST_NONE = 0, W3_HIST_RAW = 1, W3_HIST_AC = 2, W3_HIST_HUFF = 3 };
enum { W3_APM_ORDER0 = 0, W3_APM_ORDER1 = 1 };
#define W3_MAX_NODES  31
#define W3_MAX_LEAVES 16
#define W3_MAX_APM    4
#define W3_MAX_HUFF   4

typedef struct w3_node {
    uint8_t  kind;      /* W3_NODE_*                                  */
    uint8_t  bits;      /* bits_in_context   (1..32)                  */
    uint8_t  align;     /* alignment_bits    (0..7, <= bits)          */
    uint8_t  history;   /* W3_HIST_*                                  */
    uint8_t  max_bits;  /* ACHistory max_bits (0..32)                 */
    uint8_t  frozen;    /* 1 = FrozenModel wrapper                    */
    uint8_t  log_cells; /* SLOT_STATE: HashMap log_cell_count (1..24) */
    uint8_t  reserved;  /* W3_HIST_HUFF: index into spec->huff         */
    uint16_t table[8];  /* StationaryModel table, index 0 = MSB       */
} w3_node;

/* The two code tables of a HuffHistory (history/huff_history.rs:9-15): (code, len) of every byte and of every
 * partial-byte symbol (1 << bit_len | the bit_len bits seen), codes already bit-reversed as HuffHistory::new leaves
 * them (:21-25, :38-42).  Caller-supplied like StationaryModel's table: w3_huff_tables() builds them the way
 * HuffHistory::new does, or pass tables produced by the reference itself. */
typedef struct w3_huff_table {
    uint16_t code[256];     uint8_t len[256];
    uint16_t rem_code[256]; uint8_t rem_len[256];
} w3_huff_table;

/* Zero-initialise the struct before filling it (n_huff and huff are read even when no leaf uses HuffHistory:
 * n_huff > W3_MAX_HUFF, or n_huff != 0 with huff == NULL, is W3_E_INVALID; the tables themselves are only read when a
 * W3_HIST_HUFF leaf refers to them). */
typedef struct w3_model_spec {
    uint32_t n_nodes;
    w3_node  nodes[W3_MAX_NODES];
    uint32_t n_huff;               /* table sets referenced by W3_HIST_HUFF leaves (<= W3_MAX_HUFF) */
    const w3_huff_table *huff;     /* caller-owned, read-only; may be NULL when n_huff == 0          */
} w3_model_spec;

/* ---- lifecycle ------------------------------------------------------------ */
typedef struct w3_ctx w3_ctx;

/* One ctx per process and GPU; owns device workspace + a HIP stream. */
int         w3_ctx_create(int device, w3_ctx **out);
void        w3_ctx_destroy(w3_ctx *ctx);
const char *w3_strerror(int code);
const char *w3_last_error(const w3_ctx *ctx);
int         w3_abi_version(void);

/* 0 if the spec is well-formed AND implemented on the device. */
int         w3_spec_validate(const w3_model_spec *spec);

/* Options */
enum {
    W3_OPT_PATH   = 1,  /* W3_PATH_*: which device implementation encode uses      */
    W3_OPT_TIMING = 2,  /* 1 = record per-kernel hipEvent timings (w3_get_timing) */
    W3_OPT_CODER  = 3,  /* two-phase coder kernel: 0 = k_coder_x4 (default; w3_encode_submit and the half-CU variant run k_coder_x5 in its
                           place), 1 = k_coder_fast, 2 = robust k_coder only, 3 = k_coder_x2, 4 = k_coder_x3, 5 = k_coder_x5 */
    W3_OPT_ACC_LIMIT = 4, /* test hook (19..46): accumulator fill at which the fast coder hands a block back */
    W3_OPT_DEBUG_STAMPS = 5, /* diagnostic: 1 = the partitioned predict kernel sums s_memtime per phase */
    /* 6 was W3_OPT_PARTS (block ranges of ONE call pipelined on streams; measured useless, ABI v5): superseded by
       w3_encode_submit / w3_encode_wait, which pipeline successive CALLS */
    W3_OPT_VARIANT = 7, /* cross-check hook for the tests: bit mask of alternative, bit-exact implementations — 1 = Counter rounds
                           with ballots instead of returning LDS adds, 2 = 4-bit partition passes, 4 = order-2 partition from
                           scratch, 8 = CM decoder without LDS staging, 16 = no side stream, 64 = synchronous calls use the half-CU
                           kernel shapes of the submit / wait pipeline, 128 = submitted calls use the plain shapes, 256 = slot-state
                           leaves always on k_slot (lane per block, hash map in HBM), 512 = always on the sorted replay (wavefront per
                           block, no table; default: the replay below 7,000 blocks, k_slot from there on), 1024 = decode on the
                           lane-per-block kernels only (default: sixteen lanes per block where k_decode_spec applies).  0 = defaults.
                           32 = FAULT INJECTION (test hook of the sampled verification): one LDS-add round of every block returns two
                           lanes each other's value; refused (W3_E_INVALID) unless W3_OPT_VERIFY is on, so it cannot corrupt output */
    W3_OPT_SLOT_BUDGET_MB = 8, /* cap (MiB) on the device memory one batch of slot-state hash maps may take; 0 = derive from free memory */
    W3_OPT_VERIFY = 9,  /* v = 1 (default) .. 256: after every two-phase predict that used returning LDS adds — whose lane-ordered resolution
                           is measured, not documented by the ISA — max(16, nblocks * v / 256) sampled full-length blocks (at most v x 64 MiB of
                           input; the sample ROTATES from call to call, over all blocks in 256 / v calls) are predicted again with ballot rounds
                           and compared on the device; on a mismatch the call is re-encoded on the ballot path (w3_timing.n_lds_faults) and the
                           context stays there.  Cost at 1e9 B beside the coder: v = 1 +1.0 ms per 67.5 ms step, v = 2 +1.3, v = 4 +2.1.
                           RESIDUAL RISK, per call: a SYSTEMATIC change of the hardware's behaviour shows in any block and is caught by the
                           first call; a fault confined to ONE block is missed with probability 1 - v/256 (and met after at most 256 / v
                           calls); a fault that hits each block independently with probability q is missed with probability (1 - q)^S,
                           S = the sample size.  A full in-round check needs a second returning LDS atomic per add (measured
                           +4.1 % of the step for the atomics alone, and no LDS left for its shadow tables) and was not built.  Full coverage = decode the output (w3_decode_blocks_device shares no kernel with the
                           predict phase; bench.py does that for every block of its last step).  0 = off */
    W3_OPT_TUNE = 11,   /* scheduling / layout experiments (bit mask; output is identical whatever is set).  Bits 0 – 14: the submit / wait
                           pipeline's arrangements and the slot replay's shapes (HISTORY.md 2.8); 15: rank kernels with eight wavefronts per half
                           CU; 16: host-buffer copies on two streams of their own instead of the context's stream; 17 / 18: k_decode_spec with
                           the round-3 table formats / with the nibble-major ones whatever the batch size (default: by size); 19: the general k_decode_spec
                           where the instance specialised for all-raw-history models would run */
    W3_OPT_FAULT_BLOCK = 10, /* test hook, with W3_OPT_VARIANT bit 32: the one block the injected fault hits (-1 = every block, default) */
    W3_OPT_HOST_CHUNK_BLOCKS = 12 /* w3_encode_blocks: blocks per pipelined piece of a host-buffer call (0 = default: equal pieces of at most
                           4,096 blocks; tests use small values to get ragged pieces) */
};
enum { W3_PATH_AUTO = 0, W3_PATH_GENERIC = 1, W3_PATH_TWOPHASE = 2 };
int         w3_ctx_set_option(w3_ctx *ctx, int opt, int64_t value);

/* Upper bound on the concatenated block streams for n input bytes. */
size_t      w3_max_compressed_size(size_t n, size_t block_size);

/* ---- block encode / decode (host buffers) ---------------------------------
 * Replaces the bit loop of compress()/decompress() (main.rs:99-111, 127-140)
 * run once per block with a fresh model + coder: block b's stream is exactly
 * what the reference writes after its 12-byte header for a file holding only
 * that block.  Streams are byte-aligned (ACWriter::flush, io.rs:91-100) and
 * concatenated in block order; block_lens[ceil(n/block_size)] gets the sizes.
 * *out_len is set even on W3_E_NOSPACE.
 * w3_encode_blocks is PIPELINED inside the call (ABI v8): inputs of more than 4,096 blocks are cut into equal pieces of whole
 * blocks, each a w3_encode_host_submit call (below) — piece k+1's input crosses PCIe while piece k is encoded and piece k-1's
 * streams travel back; the output is byte-identical to the one-piece call's.  `in` / `out` may be pageable or pinned
 * (hipHostMalloc / hipHostRegister) memory; pinned buffers make the copies asynchronous.
 * Size limit of ONE DEVICE call (the *_device, submit / wait, stats, predict and sweep entry points): n < 2^32 - 4096 bytes,
 * W3_E_UNSUPPORTED above (a dispatch counts its work-items in 32 bits and the per-byte kernels use one per input byte).  Blocks are
 * independent: a larger device-resident input is split by the caller at block boundaries.  The HOST-buffer calls w3_encode_blocks
 * and w3_decode_blocks take any length, as the reference streams any length (main.rs:97-109): they go through in pieces of at most
 * 2 GiB of input each; w3_encode_blocks_sharded's limit applies to each device's shard.                                          */
int w3_encode_blocks(w3_ctx *ctx, const w3_model_spec *spec,
                     const uint8_t *in, size_t n, size_t block_size,
                     uint8_t *out, size_t out_cap, size_t *out_len, uint32_t *block_lens);

/* in_len = bytes readable at `in`; a length table whose sum exceeds it is rejected with W3_E_FORMAT before
 * anything is read (the reference reads through ACReader, which cannot run past its file: io.rs:23-26). */
int w3_decode_blocks(w3_ctx *ctx, const w3_model_spec *spec,
                     const uint8_t *in, size_t in_len, const uint32_t *block_lens, size_t nblocks,
                     size_t block_size, uint64_t orig_len, uint8_t *out);

/* ---- same, device-resident (no PCIe in the call) ---------------------------
 * d_* are device pointers on ctx's GPU.  `stream` is a hipStream_t; NULL = the
 * ctx's own stream, an ordinary blocking stream, i.e. ordered against work on
 * the legacy default stream like the default stream itself.  The call only
 * enqueues work and reads back one status word, so it can be timed with events
 * on that stream.
 * d_block_lens[nblocks] u32, d_total[1] u64.                                  */
int w3_encode_blocks_device(w3_ctx *ctx, const w3_model_spec *spec,
                            const uint8_t *d_in, size_t n, size_t block_size,
                            uint8_t *d_out, size_t out_cap,
                            uint32_t *d_block_lens, uint64_t *d_total, void *stream);

int w3_decode_blocks_device(w3_ctx *ctx, const w3_model_spec *spec,
                            const uint8_t *d_in, size_t in_len, const uint32_t *d_block_lens, size_t nblocks,
                            size_t block_size, uint64_t orig_len, uint8_t *d_out, void *stream);

/* ---- the same encode, asynchronous: two to four calls in flight per context --------------------------
 * The reference codes one bit at a time on one thread (main.rs:103-109); here a call is three phases with different
 * bottlenecks (predict: the store path; APM: LDS round trips; coder: one latency chain per lane on 239 of 256 CUs), and a
 * single call runs them one after the other.  w3_encode_submit only ENQUEUES: call k+1's predict phase then executes beside
 * call k's APM and coder kernels (each job has its own workspace; kernel shapes that share a CU's LDS: DESIGN.md 2.8).
 *   w3_encode_submit  arguments as w3_encode_blocks_device (d_total is required); *job receives a handle (0 .. 3).  The
 *                     input must stay valid and the outputs untouched until the job has been waited for.  `stream`: the
 *                     stream d_in was produced on (the job starts after the work enqueued there so far).  W3_E_INVALID when
 *                     w3_encode_max_in_flight(spec, n, block_size) jobs are in flight already.  Specs the predict kernels do not
 *                     cover, and specs whose slot-state leaves walk hash maps in HBM (7,000 blocks and more), run synchronously
 *                     inside the call (still completed by w3_encode_wait).
 *   w3_encode_max_in_flight  how many submitted calls of this size one context keeps in flight: 4 up to 4,096 blocks, 3 up to
 *                     12,288 — there a call's coder is a latency chain on part of an otherwise idle chip (one 64 KiB block:
 *                     17 ms), so the jobs run free, every code stage on a stream of its own, and the calls' coders overlap
 *                     (enwik8 size: 3,893 -> 9,339 MiB/s) — and 2 beyond (the ordered pair of DESIGN.md 2.8:
 *                     step k's coder beside step k+1's rank kernels; a job workspace is ~70 bytes per input byte).  Specs with
 *                     slot-state leaves: 2 (their event records are 32 bytes per input byte and leaf).
 *   w3_encode_wait    blocks until the job is complete; returns what w3_encode_blocks_device would have returned
 *                     (W3_E_NOSPACE included; d_total holds the need).  Jobs may be waited for in any order.
 * Every other entry point returns W3_E_INVALID while a job is in flight.  Output is byte-identical to the synchronous
 * call's.                                                                                                            */
int w3_encode_submit(w3_ctx *ctx, const w3_model_spec *spec,
                     const uint8_t *d_in, size_t n, size_t block_size,
                     uint8_t *d_out, size_t out_cap,
                     uint32_t *d_block_lens, uint64_t *d_total, void *stream, int *job);
int w3_encode_wait(w3_ctx *ctx, int job);
int w3_encode_max_in_flight(const w3_model_spec *spec /* NULL: a Counter-leaf model */, size_t n, size_t block_size);

/* ---- the host-buffer encode, asynchronous: calls in flight (ABI v8) -------------------------------------------
 * compress() of the reference reads a file and writes a file (main.rs:89-113): a host sees PCIe in, encode, PCIe out.  One
 * synchronous call cannot hide its own first copy in, its last coder chain (8 x block_size dependent steps per lane) and its
 * last copy out; a host with several inputs (files of a directory, main.rs:41-50; pieces of a long stream) keeps CALLS in flight:
 *   w3_encode_host_submit   arguments as w3_encode_blocks; enqueues the H2D of `in` on the context's copy-in stream and — as soon
 *                     as a device job slot is free — the encode behind it (w3_encode_submit); returns at once when `in` is pinned
 *                     memory (pageable: when HIP has staged the input).  `in`, `out`, `block_lens` must stay valid and untouched
 *                     until the job has been waited for.  *hjob receives a handle (0 .. 4).  W3_E_INVALID when
 *                     w3_encode_host_max_in_flight(spec, n, block_size) calls are in flight already.
 *   w3_encode_host_max_in_flight  = w3_encode_max_in_flight + 1: the extra call is the one whose input travels while every
 *                     device job slot is busy.
 *   w3_encode_host_wait     blocks until the job's streams and length table are in the caller's buffers; returns what
 *                     w3_encode_blocks would have returned (W3_E_NOSPACE with *out_len = the size needed included).  The jobs are
 *                     encoded in submission order; waiting for a later one first completes the earlier ones on the device (their
 *                     outputs stay on the device until they are waited for themselves).
 * Not to be mixed with w3_encode_submit jobs on the same context.  Every synchronous entry point returns W3_E_INVALID while a
 * host-buffer job is in flight.  Output byte-identical to w3_encode_blocks.                                                  */
int w3_encode_host_submit(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *in, size_t n, size_t block_size,
                          uint8_t *out, size_t out_cap, uint32_t *block_lens, int *hjob);
int w3_encode_host_wait(w3_ctx *ctx, int hjob, size_t *out_len);
int w3_encode_host_max_in_flight(const w3_model_spec *spec /* NULL: a Counter-leaf model */, size_t n, size_t block_size);

/* ---- sharding over several GPUs from ONE process (C, C++ or Rust hosts) ---------------------------
 * Blocks are independent (fresh model + coder each), so they shard with no data-path collective: context r codes the
 * contiguous block range w3_shard_range gives it (block b -> rank floor(b * world / nblocks): rank order = stream
 * order) on its own GPU and host thread; the streams land in `out` at the exclusive scan of the ranks' totals, the
 * length table is the concatenation.  Output identical to w3_encode_blocks on one context.  ctxs[] may name the same
 * device more than once (that is how the path is tested on a 1-GPU box).  Multi-PROCESS jobs (one rank per GPU,
 * bench.py) gather over RCCL instead: weath3rb0i_amd/shard.py.                                                       */
int w3_shard_range(size_t nblocks, int world, int rank, size_t *first_block, size_t *end_block);
int w3_encode_blocks_sharded(w3_ctx *const *ctxs, int n_ctx, const w3_model_spec *spec,
                             const uint8_t *in, size_t n, size_t block_size,
                             uint8_t *out, size_t out_cap, size_t *out_len, uint32_t *block_lens);

/* The same with the data RESIDENT ON THE DEVICES and the gather over xGMI — BASELINE.json's "RCCL gather over xGMI to concatenate
 * per-GPU compressed streams" for a host that is not a torch.distributed program (the reference has no counterpart: it is one
 * thread, Cargo.toml:14-15).  d_in[r] (on ctxs[r]'s device) holds shard r: the bytes of the block range w3_shard_range(nblocks,
 * n_ctx, r) of one stream, n[r] of them (a whole number of blocks for every shard but the last).  The shards are encoded
 * concurrently (one host thread per context); then ONE exchange step: an all-gather of the ranks' totals and, at the offsets of
 * their exclusive scan, grouped ncclSend / ncclRecv of the packed streams and length tables straight to the root's d_out /
 * d_block_lens (device pointers on ctxs[root]'s GPU; every peer has its own xGMI link to the root, so the transfers overlap).
 * totals[r] (host) = compressed bytes of shard r; W3_E_NOSPACE when their sum exceeds out_cap.
 * transport: W3_GATHER_AUTO = RCCL when every context has its own device, device copies otherwise; W3_GATHER_RCCL forces the
 * communicator path (one rank is allowed: that is how a 1-GPU box rehearses the RCCL calls); W3_GATHER_PEER_COPY =
 * hipMemcpyPeerAsync from the root's stream.  RCCL is resolved at the first use (dlopen "librccl.so.1"): libw3hip.so itself
 * does not link it, and W3_E_HIP with w3_last_error(ctxs[0]) = "RCCL not available ..." is returned when it is missing.
 * Output identical to w3_encode_blocks_device on one context over the concatenated shards.                               */
enum { W3_GATHER_AUTO = 0, W3_GATHER_RCCL = 1, W3_GATHER_PEER_COPY = 2 };
/* The same as a STREAM of steps (ABI v8) — the throughput form for a one-process host: every context keeps
 * w3_encode_sharded_max_in_flight(...) = min over the shards of w3_encode_max_in_flight calls in flight on its device, and a step's
 * packed streams are gathered when the step is waited for, while the devices are already coding the next steps (at 8 GPUs a 125 MB
 * shard takes 25.7 ms one call at a time and 12.0 ms with four in flight: DESIGN.md section 5).
 *   w3_encode_sharded_submit  d_in / n as above; enqueues every shard's encode (w3_encode_submit on its context, into staging buffers
 *                     of the context) and returns a step handle (0 .. 3).  Inputs must stay valid until the step has been waited
 *                     for.  Shards that w3_encode_submit runs synchronously (a tail below 8 bytes, lane-per-block specs) are coded
 *                     inside this call, one context after the other (such specs are better served by the one-shot call).
 *                     W3_E_INVALID when w3_encode_sharded_max_in_flight steps are in flight already.
 *   w3_encode_sharded_wait    completes step sjob's encodes, then the exchange step (as above: sizes all-gather + grouped ncclSend /
 *                     ncclRecv to ctxs[root]'s d_out / d_block_lens, or device copies) and returns when the streams have landed;
 *                     totals[r] = compressed bytes of shard r.  Steps may be waited for in any order.
 * Output identical to w3_encode_blocks_sharded_device's.                                                                       */
int w3_encode_sharded_submit(w3_ctx *const *ctxs, int n_ctx, const w3_model_spec *spec, const uint8_t *const *d_in, const size_t *n,
                             size_t block_size, int *sjob);
int w3_encode_sharded_wait(w3_ctx *const *ctxs, int n_ctx, int sjob, int root, uint8_t *d_out, size_t out_cap, uint32_t *d_block_lens,
                           uint64_t *totals, int transport);
int w3_encode_sharded_max_in_flight(const w3_model_spec *spec, const size_t *n, int n_ctx, size_t block_size);
/* w3_rccl_library: the library to dlopen INSTEAD of the usual sonames ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1");
 * only before the first gather or status call of the process (W3_E_INVALID afterwards).  w3_rccl_status resolves RCCL now and
 * reports: W3_OK, or W3_E_HIP with the loader's message in msg ("RCCL not available: ...").  Neither needs a device.          */
int w3_rccl_library(const char *path);
int w3_rccl_status(char *msg, size_t cap);
int w3_encode_blocks_sharded_device(w3_ctx *const *ctxs, int n_ctx, const w3_model_spec *spec,
                                    const uint8_t *const *d_in, const size_t *n, size_t block_size, int root,
                                    uint8_t *d_out, size_t out_cap, uint32_t *d_block_lens,
                                    uint64_t *totals, int transport);

/* ---- ACStats, the counting sink (helpers.rs:60-90) -------------------------------
 * Every figure the reference publishes is `csize = bits / 8` from this sink (bin/ordern/main.rs:66-80): write_bit counts
 * 1 + the pending parity bits it resolves, flush adds nothing (:87-89).  block_bits[b] = that count for block b coded
 * alone (u32: blocks below 2^28 bytes); csize of a block = block_bits[b] / 8 (:70-73).  Same kernels as the encode,
 * without the pack; nothing is copied out but the counts.                                                          */
int w3_encode_stats(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *in, size_t n, size_t block_size, uint32_t *block_bits);
int w3_encode_stats_device(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *d_in, size_t n, size_t block_size,
                           uint32_t *d_block_bits, void *stream);

/* ---- parameter sweep (bin/ordern/main.rs:9-80) ----------------------------------
 * OrderN::new(bits[c], aligns[c]) for every configuration c, each through the counting sink on every block, ALL IN ONE
 * LAUNCH (lanes = configurations x blocks; batches only when the Counter tables exceed the device memory).
 * block_bits (host memory) = [ncfg][nblocks] bit counts, configuration-major.  The reference runs the configurations one
 * after the other on one thread; the driver around this call (weath3rb0i_amd/sweep.py) prints its lines.            */
int w3_sweep_ordern(w3_ctx *ctx, const uint8_t *in, size_t n, size_t block_size, const uint8_t *bits, const uint8_t *aligns,
                    size_t ncfg, uint32_t *block_bits);
int w3_sweep_ordern_device(w3_ctx *ctx, const uint8_t *d_in, size_t n, size_t block_size, const uint8_t *bits, const uint8_t *aligns,
                           size_t ncfg, uint32_t *block_bits);

/* ---- context statistics export (README.md:9: "output stats from contexts for use by external neural nets") --------
 * The Counter table (`stats`, models/ordern.rs:5 / ordern_entropy.rs:6) of a one-leaf adaptive model after it has seen
 * `in` as ONE stream, i.e. the state the reference's model is in when compress() returns: counters[ctx] = n0 | n1 << 16
 * (models/counter.rs:4-6) for ctx in 0 .. 2^bits_in_context.  bits_in_context <= 28, n <= 2^28.                       */
int w3_export_counters(w3_ctx *ctx, const w3_model_spec *spec, const uint8_t *in, size_t n, uint32_t *counters);

/* ---- the reference's whole-file container ----------------------------------
 * compress()/decompress() of main.rs:89-144: b"w30i" + u64 BE length + ONE
 * stream.  One serial chain => one GPU lane; provided for format parity.
 * Inputs above 2^28 bytes are refused with W3_E_UNSUPPORTED (w3_last_error names the block container as the route:
 * a single lane codes 2^28 bytes in about ten minutes; the reference's format has no blocks to code in parallel). */
int w3_compress_stream(w3_ctx *ctx, const w3_model_spec *spec,
                       const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, size_t *out_len);
int w3_decompress_stream(w3_ctx *ctx, const w3_model_spec *spec,
                         const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len);

/* ---- per-step probabilities (Model::predict for every bit of every block) --
 * models/mod.rs:12-15 evaluated by the device predict phase; p_out[8*n] u16.
 * Lets a test drive the model surface without the coder.                      */
int w3_predict_blocks(w3_ctx *ctx, const w3_model_spec *spec,
                      const uint8_t *in, size_t n, size_t block_size, uint16_t *p_out);

/* ---- StationaryModel::new(buf) (models/ac_hash/stationary.rs:14-34) ---------
 * Host-side table preparation for ACHistory: 8 Counters by bit position walked
 * over `buf`, table[i] = p().  Model construction, not the hot path.          */
int w3_stationary_table(const uint8_t *buf, size_t n, uint16_t table[8]);

/* ---- HuffHistory::new(buf, huff_size, rem_huff_size) (history/huff_history.rs:17-55) --------
 * Host-side table preparation: byte histogram (helpers.rs:30-36) -> length-limited Huffman code lengths
 * (entropy_coding/package_merge.rs:1-84) -> canonical codes (:87-117), bit-reversed; the same for the 255 partial-byte
 * symbols.  The reference sorts with sort_unstable_by (:9, :92): among EQUAL counts / lengths its order is an
 * implementation detail of Rust's unstable sort; this implementation takes them in ascending symbol order (the order the
 * reference's own tests show for small inputs).  REFERENCE-IDENTICAL TABLES for a histogram with tied counts or lengths
 * therefore come from the crate itself: a Rust host passes HuffHistory's own tables through w3_model_spec.huff (from_tables in
 * INTEGRATION.md); Rust's unstable sort is not re-derived here.  Model construction, not the hot path.
 * Returns W3_E_INVALID for the reference's panics (no symbols, max length > 32 or too small for the alphabet).      */
int w3_huff_tables(const uint8_t *buf, size_t n, uint8_t huff_size, uint8_t rem_huff_size, w3_huff_table *out);

/* ---- read-only tables of the CM kernels (host-side; for known-answer tests) -----------
 * w3_state_table: NaiveStateTable (state_table/naive.rs:7-115) as 3963 rows of
 * {prob, next[0], next[1]} — the rows of docs/state_table/state_table.csv.
 * w3_stretch_squash: the build-defined logistic pair, stretch[4096] (index p>>4) and
 * squash[4095] (index d+2047).                                                          */
#define W3_STATE_TABLE_SIZE 3963
int w3_state_table(uint16_t *out /* [3963*3] */);
int w3_stretch_squash(int16_t *stretch /* [4096] */, uint16_t *squash /* [4095] */);

/* ---- device self-test ---------------------------------------------------------
 * Exhaustively compares the kernels' division-free Counter::p with the literal
 * u64 formula (models/counter.rs:13-18) for all 2^32 (c0,c1) states, on the GPU. */
int w3_selftest_counter_p(w3_ctx *ctx, uint64_t *mismatches);

/* Diagnostic: per-phase s_memtime sums of the last partitioned predict kernel (W3_OPT_DEBUG_STAMPS=1):
 * [0] digit histograms, [1] first partition pass, [2] remaining passes, [3] rank loop, [7] blocks.       */
int w3_debug_get_stamps(w3_ctx *ctx, uint64_t out[8]);

/* ---- timing of the last encode call (W3_OPT_TIMING=1) ----------------------- */
typedef struct w3_timing {
    float    predict_ms;   /* context + rank + Counter::p kernels            */
    float    coder_ms;     /* lane-per-block arithmetic coder kernel         */
    float    pack_ms;      /* scan + compaction of the block streams         */
    float    generic_ms;   /* fused lane-per-block kernel (generic path)     */
    float    total_ms;
    uint32_t path;         /* W3_PATH_* actually taken                       */
    uint32_t n_coder_launches;
    uint64_t coder_bytes;  /* algorithmic HBM bytes of the coder launches    */
    uint64_t predict_bytes;
    uint32_t n_recoded_blocks; /* blocks the fast coder handed to the robust coder */
    float    apm_ms;       /* APM stage kernels of the two-phase path (k_apm0 / k_apm1, + k_mix / k_partition they need) */
    float    slot_ms;      /* slot-state leaves: table zero-fill + k_slot launches (also inside predict_ms) */
    uint32_t n_slot_launches; /* k_slot launches (block batches sized to the device memory budget) */
    float    achash_ms;    /* ACHistory key kernels (k_achash_lut + k_achash; also inside predict_ms) */
    uint32_t n_parts;      /* pieces the call was cut into: 1 for every device-resident call, >= 1 for w3_encode_blocks (ABI v8: the
                              times and byte counts of this struct are then sums over the pieces) */
    uint32_t n_lds_faults; /* wavefronts of the sampled verification whose streams differed (W3_OPT_VERIFY); > 0: the call was
                              re-encoded with ballot rounds */
    /* launch durations of the predict phase's kernels (they may overlap in time: side stream, or another job's kernels) */
    uint32_t n_wide;       /* wide Counter leaves (H = 16 / 24) of the spec, in leaf order (at most 4 are timed) */
    float    part_ms[4];   /* k_partition8 / k_partition pass of wide leaf w */
    float    rank_ms[4];   /* k_rank_sorted of wide leaf w */
    float    small_ms;     /* k_predict_small launches of the time-ordered Counter leaves (H <= 8, raw history), together */
} w3_timing;
int w3_get_timing(const w3_ctx *ctx, w3_timing *out);

#ifdef __cplusplus
}
#endif
#endif
