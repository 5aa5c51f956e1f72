/*
 * w3_oracle.h — CPU ORACLE for the weath3rb0i hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement (written by reading, not copied) of the
 * reference's per-bit context-model prediction + binary arithmetic coding
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product (weath3rb0i_amd / libw3hip.so) never links,
 * imports or calls anything in this directory.
 *
 * Parity status: PINNED for the coder + bit I/O (reference unit-test vectors,
 * entropy_coding/io.rs:107-227, entropy_coding/arithmetic_coder_tests.rs:44-161),
 * the state table (docs/state_table CSV digests) and Slot indexing
 * (docs/hashslots.md:69-133).  The models (Counter, Order*, History,
 * BestOfTwo) have no reference test vectors: they are restated by reading and
 * cross-checked against the SURVEY §8(c) digests ("parity unpinned" for the
 * end-to-end model streams).
 *
 * All citations are path:line under /root/reference/src unless stated.
 */
#ifndef W3_ORACLE_H
#define W3_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Bit sinks: the three ACWrite impls of the reference                 */
/*   W3O_SINK_BYTES   = ACWriter<W>      entropy_coding/io.rs:52-101   */
/*   W3O_SINK_STATS   = ACStats          helpers.rs:60-90              */
/*   W3O_SINK_ENTROPY = EntropyWriter    history/ac_history.rs:50-89   */
/* ------------------------------------------------------------------ */
enum { W3O_SINK_BYTES = 0, W3O_SINK_STATS = 1, W3O_SINK_ENTROPY = 2 };

typedef struct w3o_sink {
    int kind;
    /* BYTES */
    uint8_t *buf; size_t len, cap; int owns;
    uint8_t  acc, idx;
    uint64_t rev_bits;
    /* STATS */
    uint64_t bit_count;
    /* ENTROPY */
    uint32_t state; uint8_t max_bits, eidx; uint16_t erev;
} w3o_sink;

void   w3o_sink_init_bytes(w3o_sink *s);               /* growable, owned */
void   w3o_sink_init_stats(w3o_sink *s);
void   w3o_sink_init_entropy(w3o_sink *s, uint8_t max_bits);
void   w3o_sink_free(w3o_sink *s);
void   w3o_sink_inc_parity(w3o_sink *s);
int    w3o_sink_write_bit(w3o_sink *s, uint8_t bit);   /* 0 ok, -1 = Err (entropy sink full) */
int    w3o_sink_flush(w3o_sink *s, uint32_t state);
uint64_t w3o_stats_result(const w3o_sink *s);          /* helpers.rs:70-73 */

/* ACReader  entropy_coding/io.rs:7-49 */
typedef struct w3o_reader { const uint8_t *buf; size_t len, pos; uint8_t cur, mask; } w3o_reader;
void     w3o_reader_init(w3o_reader *r, const uint8_t *buf, size_t len);
uint8_t  w3o_reader_read_bit(w3o_reader *r);
uint32_t w3o_reader_read_u32(w3o_reader *r);

/* ArithmeticCoder  entropy_coding/arithmetic_coder.rs:13-119 */
typedef struct w3o_ac { uint32_t x1, x2, x; } w3o_ac;
void    w3o_ac_new_coder(w3o_ac *ac);
int     w3o_ac_encode(w3o_ac *ac, uint8_t bit, uint16_t prob, w3o_sink *io);
int     w3o_ac_flush(w3o_ac *ac, w3o_sink *io);
void    w3o_ac_new_decoder(w3o_ac *ac, w3o_reader *r);
uint8_t w3o_ac_decode(w3o_ac *ac, uint16_t prob, w3o_reader *r);

/* Counter  models/counter.rs:4-26 */
typedef struct w3o_counter { uint16_t data[2]; } w3o_counter;
uint16_t w3o_counter_p(const w3o_counter *c);
void     w3o_counter_update(w3o_counter *c, uint8_t bit);

/* StationaryModel  models/ac_hash/stationary.rs:8-58 */
typedef struct w3o_stationary { uint16_t table[8]; uint8_t alignment; } w3o_stationary;
void w3o_stationary_new(w3o_stationary *m, const uint8_t *buf, size_t n);
void w3o_stationary_from_table(w3o_stationary *m, const uint16_t table[8]);
void w3o_stationary_for_book1(w3o_stationary *m);
void w3o_stationary_for_enwik7(w3o_stationary *m);

/* Length-limited Huffman (entropy_coding/package_merge.rs): host-side table preparation of HuffHistory.
 * The reference sorts with sort_unstable_by (package_merge.rs:9, :92): the order of EQUAL keys is an implementation detail
 * of Rust's unstable sort (insertion sort = stable up to 20 elements, pattern-defeating quicksort beyond).  This
 * restatement sorts STABLY (ties in ascending symbol order): identical whenever the reference's result does not depend
 * on the tie order — all 12 reference tests (package_merge.rs:127-267) are of that kind, see tests/test_oracle_kats.py —
 * and a documented choice otherwise.  Return codes mirror the reference's asserts (:13-18). */
enum { W3O_PM_OK = 0, W3O_PM_NO_SYMBOLS = -1, W3O_PM_MAX_LEN_TOO_BIG = -2, W3O_PM_MAX_LEN_TOO_SMALL = -3 };
int  w3o_package_merge(const uint32_t *counts, size_t n, uint8_t max_len, uint8_t *code_lens /* [n] */);      /* :1-29, :34-84 */
void w3o_canonical(const uint8_t *code_lens, size_t n, uint16_t *codes /* [n] */, uint8_t *lens /* [n] */);    /* :87-117 */

/* The two code tables of a HuffHistory (history/huff_history.rs:9-15): (code, len) per byte and per partial-byte symbol
 * (1 << bit_len | top bits), codes bit-reversed as at :21-25 / :38-42. */
typedef struct w3o_huff_tables { uint16_t code[256]; uint8_t len[256]; uint16_t rem_code[256]; uint8_t rem_len[256]; } w3o_huff_tables;
int  w3o_huff_tables_new(const uint8_t *buf, size_t n, uint8_t huff_size, uint8_t rem_huff_size, w3o_huff_tables *out);   /* HuffHistory::new :17-55 */

/* History  history/{mod,raw_history,ac_history,huff_history}.rs */
enum { W3O_HIST_RAW = 0, W3O_HIST_AC = 1, W3O_HIST_HUFF = 2 };
typedef struct w3o_history {
    int kind;
    uint32_t raw_bits;                       /* RawHistory */
    uint64_t pos, bits; uint8_t max_bits;    /* ACHistory (pos, bits also HuffHistory) */
    w3o_stationary model;
    uint32_t compressed_bits;                /* HuffHistory */
    w3o_huff_tables huff;
} w3o_history;
void     w3o_history_raw(w3o_history *h);
void     w3o_history_ac(w3o_history *h, uint8_t max_bits, const w3o_stationary *m);
void     w3o_history_huff(w3o_history *h, const w3o_huff_tables *t);
void     w3o_history_update(w3o_history *h, uint8_t bit);
uint32_t w3o_history_hash(w3o_history *h);

/* ACHistoryCached  history/ac_history_cached.rs:9-76: the memoised ACHistory, restated with its two-level memo keyed by
 * (bits & mask, alignment, level).  The product has no node for it (its hash() equals ACHistory's: weath3rb0i_amd/models.py);
 * this literal form exists so that the equality is a TEST RESULT (tests/test_oracle_kats.py), not an argument.
 * counts: [level-0 hits, level-1 hits, full runs, memo entries]. */
typedef struct w3o_achc w3o_achc;
w3o_achc *w3o_achc_new(uint8_t max_bits, const w3o_stationary *m, uint8_t cache_size);   /* :19-28 */
void      w3o_achc_free(w3o_achc *h);
void      w3o_achc_update(w3o_achc *h, uint8_t bit);   /* :32-35 */
uint32_t  w3o_achc_hash(w3o_achc *h);                  /* :37-76 */
void      w3o_achc_counts(const w3o_achc *h, uint64_t out[4]);

/* Models: trait Model (models/mod.rs:12-15) as an opaque tree */
typedef struct w3o_model w3o_model;
w3o_model *w3o_order0(void);                                   /* models/order0.rs */
w3o_model *w3o_order1(void);                                   /* models/order1.rs */
w3o_model *w3o_ordern(uint8_t bits, uint8_t align);            /* models/ordern.rs */
w3o_model *w3o_ordern_entropy(uint8_t bits, uint8_t align, const w3o_history *h); /* models/ordern_entropy.rs */
w3o_model *w3o_frozen(w3o_model *adaptive);                    /* models/frozen.rs  (takes ownership) */
w3o_model *w3o_best_of_two(w3o_model *m1, w3o_model *m2);      /* models/mod.rs:42-75 (takes ownership) */
/* ---- BUILD-DEFINED models (SURVEY §8 A19 ii-v; no reference counterpart, "parity unpinned") ----
 * w3o_slot_model: the state-table CM leaf hashslots.md describes, wiring NaiveStateTable (A15)
 *   to the Cell/Slot hashmap (A16): one cell touch per nibble, 15 12-bit states per slot.
 * w3o_apm: adaptive probability map over stretch(p) with 33 interpolated buckets per row
 *   ("APM mixers", README.md:10).  See DESIGN.md §2.4 for the integer definitions. */
w3o_model *w3o_slot_model(uint8_t order, uint8_t log_cells);
enum { W3O_APM_ORDER0 = 0, W3O_APM_ORDER1 = 1 };
w3o_model *w3o_apm(w3o_model *input, uint8_t ctx_kind, uint8_t rate);   /* takes ownership */
uint16_t   w3o_squash(int d);            /* d in [-2047,2047] -> P(1)*2^16 in [1,65535] */
int        w3o_stretch(uint16_t p);      /* inverse on p>>4, in [-2047,2047]            */
uint64_t   w3o_slot_hash(uint8_t order, uint64_t hist_bytes, int second, uint32_t hi_nib);
uint8_t    w3o_st_conf(uint16_t state);  /* observation count proxy used by the slot replacement policy */
w3o_model *w3o_model_clone_fresh(const w3o_model *m);          /* same structure, initial state */
void       w3o_model_reset(w3o_model *m);
void       w3o_model_free(w3o_model *m);
uint16_t   w3o_model_predict(const w3o_model *m);
void       w3o_model_update(w3o_model *m, uint8_t bit);        /* Model::update: adapt then update */
uint16_t   w3o_opinion_mix(uint16_t p1, uint16_t p2);          /* mixers/opinion_mixer2.rs:5-10 */

/* The bit loop  main.rs:103-111 (no header).  Returns malloc'd stream. */
uint8_t *w3o_encode_stream(w3o_model *m, const uint8_t *in, size_t n, size_t *out_len);
/* Same loop into the counting sink (bin/order0/main.rs:13-25): returns csize = bits/8 */
uint64_t w3o_encode_stats(w3o_model *m, const uint8_t *in, size_t n);
uint64_t w3o_encode_stats_bits(w3o_model *m, const uint8_t *in, size_t n);   /* the raw bit count (csize = bits / 8) */
/* decode loop  main.rs:131-140 */
void w3o_decode_stream(w3o_model *m, const uint8_t *in, size_t in_len, uint8_t *out, size_t n);
/* the per-step probabilities (debug aid for the two-phase GPU path) */
void w3o_predict_all(w3o_model *m, const uint8_t *in, size_t n, uint16_t *p_out);

/* Reference container  main.rs:14-15,89-144 : "w30i" + u64 BE len + stream */
uint8_t *w3o_compress_container(w3o_model *m, const uint8_t *in, size_t n, size_t *out_len);
int      w3o_decompress_container(w3o_model *m, const uint8_t *in, size_t in_len, uint8_t **out, size_t *out_len);

/* Block mode (build-defined): every block is the stream the
 * reference would emit for a file holding only that block (fresh model+coder).
 * out must hold the concatenation; block_lens[ceil(n/bs)] gets per-block sizes.
 * nthreads>1 spreads blocks over pthreads (cpu_baseline "all cores" leg). */
int w3o_encode_blocks(const w3o_model *proto, const uint8_t *in, size_t n, size_t block_size,
                      uint8_t *out, size_t out_cap, size_t *out_len, uint32_t *block_lens, int nthreads);
int w3o_decode_blocks(const w3o_model *proto, const uint8_t *in, const uint32_t *block_lens,
                      size_t nblocks, size_t block_size, uint64_t orig_len, uint8_t *out, int nthreads);

/* NaiveStateTable  state_table/naive.rs:7-115, trait state_table/mod.rs:3-23 */
#define W3O_ST_SIZE 3963
#define W3O_ST_AUX  990
typedef struct w3o_state_entry { uint16_t prob, next[2]; } w3o_state_entry;
const w3o_state_entry *w3o_state_table(void);       /* 3963 entries */
const w3o_state_entry *w3o_state_table_aux(void);   /*  990 entries */
uint16_t w3o_st_next(uint16_t state, uint8_t bit);
uint16_t w3o_st_p(uint16_t state);
void     w3o_st_next4(const uint16_t s[4], uint8_t nib, uint16_t out[4]);
void     w3o_st_p4(const uint16_t s[4], uint16_t out[4]);

/* HashMap / Cell / Slot  hashmap.rs:1-129 */
typedef struct w3o_cell { uint8_t hashes[6]; uint8_t slots[90]; } w3o_cell;
uint32_t w3o_hashmap_log_cell_count(size_t size_bytes);                 /* hashmap.rs:7-10 */
uint64_t w3o_hashmap_cell_index(uint64_t hash, uint32_t log_cell_count);/* hashmap.rs:25-28 */
uint8_t  w3o_cell_get_slot(const w3o_cell *c, uint64_t hash);           /* hashmap.rs:42-71 -> slot id */
void     w3o_slot_get_idx(uint8_t id, uint8_t bit_id, uint8_t nib_ctx, uint32_t *abs_idx, int *parity);
uint16_t w3o_slot_get_state(const w3o_cell *c, uint8_t id, uint8_t bit_id, uint8_t nib_ctx);
void     w3o_slot_set_state(w3o_cell *c, uint8_t id, uint8_t bit_id, uint8_t nib_ctx, uint16_t st);
void     w3o_slot_get_nib(const w3o_cell *c, uint8_t id, uint8_t nib, uint16_t out[4]);
void     w3o_slot_set_nib(w3o_cell *c, uint8_t id, uint8_t nib, const uint16_t st[4]);

#ifdef __cplusplus
}
#endif
#endif
