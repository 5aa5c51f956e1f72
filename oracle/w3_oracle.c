/*
 * w3_oracle.c — CPU ORACLE (test infrastructure only; see w3_oracle.h).
 * Plain-C restatement of the reference hot path, written from the reference's
 * source by reading.  Every function cites the reference file:line it follows
 * (paths under /root/reference/src).  Parity pinning: see header + DESIGN.md.
 */
#include "w3_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ================================================================== */
/* Bit sinks                                                           */
/* ================================================================== */

void w3o_sink_init_bytes(w3o_sink *s) {
    memset(s, 0, sizeof *s);
    s->kind = W3O_SINK_BYTES;
    s->cap = 1 << 12;
    s->buf = (uint8_t *)malloc(s->cap);
    s->owns = 1;
}
void w3o_sink_init_stats(w3o_sink *s) { memset(s, 0, sizeof *s); s->kind = W3O_SINK_STATS; }
void w3o_sink_init_entropy(w3o_sink *s, uint8_t max_bits) {
    memset(s, 0, sizeof *s);
    s->kind = W3O_SINK_ENTROPY;
    s->max_bits = max_bits;
}
void w3o_sink_free(w3o_sink *s) {
    if (s->owns && s->buf) free(s->buf);
    s->buf = NULL;
}

static void sink_push_byte(w3o_sink *s, uint8_t b) {
    if (s->len == s->cap) {
        s->cap *= 2;
        s->buf = (uint8_t *)realloc(s->buf, s->cap);
    }
    s->buf[s->len++] = b;
}

/* io.rs:66-68, helpers.rs:77-79, ac_history.rs:82-84 */
void w3o_sink_inc_parity(w3o_sink *s) {
    switch (s->kind) {
    case W3O_SINK_BYTES: s->rev_bits += 1; break;
    case W3O_SINK_STATS: s->rev_bits += 1; break;
    default:             s->erev += 1;     break;
    }
}

/* io.rs:74-80 (write_bit_raw closure): MSB-first packing, byte out when idx wraps */
static void bytes_write_raw(w3o_sink *s, uint8_t bit) {
    s->acc = (uint8_t)((s->acc << 1) | bit);
    s->idx = (uint8_t)((s->idx + 1) % 8);
    if (s->idx == 0) sink_push_byte(s, s->acc);
}

/* ac_history.rs:63-71 (write_bit_raw closure): Err once max_bits are in */
static int entropy_write_raw(w3o_sink *s, uint8_t bit) {
    if (s->eidx == s->max_bits) return -1;
    s->state = (s->state >> 1) | ((uint32_t)bit << 31);
    s->eidx += 1;
    return 0;
}

int w3o_sink_write_bit(w3o_sink *s, uint8_t bit) {
    switch (s->kind) {
    case W3O_SINK_BYTES: /* io.rs:70-89 */
        bytes_write_raw(s, bit);
        while (s->rev_bits > 0) {
            s->rev_bits -= 1;
            bytes_write_raw(s, bit ^ 1);
        }
        return 0;
    case W3O_SINK_STATS: /* helpers.rs:81-85 */
        s->bit_count += 1 + s->rev_bits;
        s->rev_bits = 0;
        return 0;
    default: /* ac_history.rs:58-80 */
        if (entropy_write_raw(s, bit)) return -1;
        while (s->erev > 0) {
            s->erev -= 1;
            if (entropy_write_raw(s, bit ^ 1)) return -1;
        }
        return 0;
    }
}

int w3o_sink_flush(w3o_sink *s, uint32_t state) {
    switch (s->kind) {
    case W3O_SINK_BYTES: /* io.rs:91-100: at least one bit, then pad to a byte with state's MSBs */
        w3o_sink_write_bit(s, (uint8_t)(state >> 31));
        state <<= 1;
        while (s->idx > 0) {
            w3o_sink_write_bit(s, (uint8_t)(state >> 31));
            state <<= 1;
        }
        return 0;
    case W3O_SINK_STATS: /* helpers.rs:87-89 */
        return 0;
    default: /* ac_history.rs:86-88 unimplemented!() */
        return -1;
    }
}

uint64_t w3o_stats_result(const w3o_sink *s) { return s->bit_count / 8; }

/* ================================================================== */
/* ACReader  io.rs:7-49                                                */
/* ================================================================== */

void w3o_reader_init(w3o_reader *r, const uint8_t *buf, size_t len) {
    r->buf = buf; r->len = len; r->pos = 0; r->cur = 0; r->mask = 0;
}
static uint8_t reader_read_byte(w3o_reader *r) { /* io.rs:18-27: 0 past EOF */
    if (r->pos < r->len) return r->buf[r->pos++];
    return 0;
}
uint8_t w3o_reader_read_bit(w3o_reader *r) { /* io.rs:31-38 */
    r->mask >>= 1;
    if (r->mask == 0) {
        r->cur = reader_read_byte(r);
        r->mask = 1 << 7;
    }
    return (r->cur & r->mask) > 0;
}
uint32_t w3o_reader_read_u32(w3o_reader *r) { /* io.rs:40-48: 4 bytes BE, zero padded */
    uint32_t b0 = reader_read_byte(r), b1 = reader_read_byte(r);
    uint32_t b2 = reader_read_byte(r), b3 = reader_read_byte(r);
    return (b0 << 24) | (b1 << 16) | (b2 << 8) | b3;
}

/* ================================================================== */
/* ArithmeticCoder  arithmetic_coder.rs                                */
/* ================================================================== */

#define PREC_SHIFT 31u
#define Q1 0x40000000u
#define Q2 0x80000000u
#define Q3 0xC0000000u
#define RLO_MOD 0x7FFFFFFFu
#define RHI_MOD 0x80000001u

/* arithmetic_coder.rs:109-119 */
static inline uint32_t lerp(uint32_t x1, uint32_t x2, uint16_t prob) {
    uint64_t p = prob == 0 ? 1 : ((uint64_t)prob << 16);
    uint64_t range = (uint64_t)(x2 - x1);
    uint64_t lerped = (range * p) >> 32;
    return x1 + (uint32_t)lerped;
}

void w3o_ac_new_coder(w3o_ac *ac) { ac->x1 = 0; ac->x2 = 0xFFFFFFFFu; ac->x = 0; } /* :37-39 */

int w3o_ac_encode(w3o_ac *ac, uint8_t bit, uint16_t prob, w3o_sink *io) { /* :41-65 */
    uint32_t xmid = lerp(ac->x1, ac->x2, prob);
    if (bit == 0) ac->x1 = xmid + 1; else ac->x2 = xmid;

    while (((ac->x1 ^ ac->x2) >> PREC_SHIFT) == 0) {
        if (w3o_sink_write_bit(io, (uint8_t)(ac->x1 >> PREC_SHIFT))) return -1;
        ac->x1 <<= 1;
        ac->x2 = (ac->x2 << 1) | 1;
    }
    while (ac->x1 >= Q1 && ac->x2 < Q3) {
        w3o_sink_inc_parity(io);
        ac->x1 = (ac->x1 << 1) & RLO_MOD;
        ac->x2 = (ac->x2 << 1) | RHI_MOD;
    }
    return 0;
}

int w3o_ac_flush(w3o_ac *ac, w3o_sink *io) { return w3o_sink_flush(io, ac->x2); } /* :67-71 */

void w3o_ac_new_decoder(w3o_ac *ac, w3o_reader *r) { /* :75-78 */
    ac->x = w3o_reader_read_u32(r);
    ac->x1 = 0; ac->x2 = 0xFFFFFFFFu;
}

uint8_t w3o_ac_decode(w3o_ac *ac, uint16_t prob, w3o_reader *r) { /* :80-105 */
    uint32_t xmid = lerp(ac->x1, ac->x2, prob);
    uint8_t bit = ac->x <= xmid;
    if (bit == 0) ac->x1 = xmid + 1; else ac->x2 = xmid;

    while (((ac->x1 ^ ac->x2) >> PREC_SHIFT) == 0) {
        ac->x1 <<= 1;
        ac->x2 = (ac->x2 << 1) | 1;
        ac->x = (ac->x << 1) | w3o_reader_read_bit(r);
    }
    while (ac->x1 >= Q1 && ac->x2 < Q3) {
        ac->x1 = (ac->x1 << 1) & RLO_MOD;
        ac->x2 = (ac->x2 << 1) | RHI_MOD;
        ac->x = ((ac->x << 1) ^ Q2) | w3o_reader_read_bit(r);
    }
    return bit;
}

/* ================================================================== */
/* Counter  models/counter.rs                                          */
/* ================================================================== */

uint16_t w3o_counter_p(const w3o_counter *c) { /* :13-18 */
    uint64_t c0 = c->data[0], c1 = c->data[1];
    uint64_t p = ((uint64_t)1 << 17) * (c1 + 1) / (c0 + c1 + 2);
    return (uint16_t)((p >> 1) + (p & 1));
}
void w3o_counter_update(w3o_counter *c, uint8_t bit) { /* :20-26 */
    c->data[bit] += 1;
    if (c->data[bit] == 0xFFFF) {
        c->data[0] = (uint16_t)((c->data[0] >> 1) + (c->data[0] & 1));
        c->data[1] = (uint16_t)((c->data[1] >> 1) + (c->data[1] & 1));
    }
}

/* ================================================================== */
/* StationaryModel  models/ac_hash/stationary.rs                       */
/* ================================================================== */

void w3o_stationary_new(w3o_stationary *m, const uint8_t *buf, size_t n) { /* :14-34 */
    w3o_counter model[8];
    memset(model, 0, sizeof model);
    for (size_t k = 0; k < n; k++) {
        unsigned i = 7;
        for (int s = 7; s >= 0; s--) { /* unroll_for!: MSB first, macros.rs:1-21 */
            i = (i + 1) & 7;
            w3o_counter_update(&model[i], (buf[k] >> s) & 1);
        }
    }
    for (int i = 0; i < 8; i++) m->table[i] = w3o_counter_p(&model[i]);
    m->alignment = 0;
}
void w3o_stationary_from_table(w3o_stationary *m, const uint16_t table[8]) { /* :36-38 */
    memcpy(m->table, table, sizeof m->table);
    m->alignment = 0;
}
void w3o_stationary_for_book1(w3o_stationary *m) { /* :40-42 */
    static const uint16_t t[8] = {1, 50188, 62497, 15819, 22545, 31499, 22988, 29616};
    w3o_stationary_from_table(m, t);
}
void w3o_stationary_for_enwik7(w3o_stationary *m) { /* :44-46 */
    static const uint16_t t[8] = {752, 50314, 58928, 21421, 24680, 30788, 24297, 32530};
    w3o_stationary_from_table(m, t);
}
static inline void stationary_align(w3o_stationary *m, uint8_t a) { m->alignment = a; } /* :50-52 */
static inline uint16_t stationary_predict(w3o_stationary *m) { /* :54-57: walks backwards */
    m->alignment = (uint8_t)((m->alignment + 7) & 7);
    return m->table[m->alignment];
}

/* ================================================================== */
/* Length-limited Huffman  entropy_coding/package_merge.rs             */
/* ================================================================== */

typedef struct { uint32_t sym, key; } pm_item;
static void pm_stable_sort(pm_item *a, size_t n) {   /* insertion sort: stable, n <= 256 */
    for (size_t i = 1; i < n; i++) {
        pm_item x = a[i];
        size_t j = i;
        while (j > 0 && a[j - 1].key > x.key) { a[j] = a[j - 1]; j--; }
        a[j] = x;
    }
}

/* package_merge_sorted (:34-84): `a` ascending; returns the code length of every element */
static void pm_sorted(const uint32_t *a, size_t n, uint8_t max_len, uint8_t *code_lens) {
    if (n == 1) { code_lens[0] = 0; return; }                    /* 2n-2 = 0 relevant symbols: every length stays 0 */
    const size_t cap = 2 * n - 1;
    uint32_t *depths = (uint32_t *)calloc(cap, sizeof(uint32_t));
    uint64_t *prev = (uint64_t *)malloc(cap * sizeof(uint64_t)), *curr = (uint64_t *)malloc(cap * sizeof(uint64_t));
    size_t nprev = n, ncurr = 0;
    for (size_t i = 0; i < n; i++) prev[i] = a[i];
    for (unsigned depth = 1; depth < max_len; depth++) {         /* :40-63 */
        const uint32_t mask = 1u << depth;
        size_t si = 0, pi = 0;
        const size_t npk = nprev / 2;                            /* chunks_exact(2) */
        ncurr = 0;
        for (;;) {
            int is_package;
            if (pi >= npk && si >= n) break;
            if (pi >= npk) is_package = 0;
            else if (si >= n) is_package = 1;
            else is_package = (prev[2 * pi] + prev[2 * pi + 1]) <= a[si];
            if (is_package) { depths[ncurr] |= mask; curr[ncurr++] = prev[2 * pi] + prev[2 * pi + 1]; pi++; }
            else curr[ncurr++] = a[si++];
        }
        uint64_t *t = prev; prev = curr; curr = t;
        nprev = ncurr;
    }
    memset(code_lens, 0, n);
    size_t relevant = 2 * n - 2;                                  /* :66-82 */
    for (int depth = (int)max_len - 1; depth >= 0; depth--) {
        if (relevant == 0) break;
        const uint32_t mask = 1u << depth;
        size_t sym = 0;
        for (size_t i = 0; i < relevant && i < cap; i++)
            if ((depths[i] & mask) == 0) { code_lens[sym] += 1; sym++; }
        relevant = (relevant - sym) * 2;
    }
    free(depths); free(prev); free(curr);
}

int w3o_package_merge(const uint32_t *counts, size_t n, uint8_t max_len, uint8_t *code_lens) { /* :1-29 */
    pm_item *it = (pm_item *)malloc((n ? n : 1) * sizeof(pm_item));
    size_t m = 0;
    for (size_t i = 0; i < n; i++)
        if (counts[i] != 0) { it[m].sym = (uint32_t)i; it[m].key = counts[i]; m++; }
    pm_stable_sort(it, m);                                        /* :9 sort_unstable_by: ties in ascending symbol order here */
    int rc = W3O_PM_OK;
    if (m == 0) rc = W3O_PM_NO_SYMBOLS;                           /* :12 */
    else if (max_len > 32) rc = W3O_PM_MAX_LEN_TOO_BIG;           /* :13 */
    else if (max_len < 32 && m > ((size_t)1 << max_len)) rc = W3O_PM_MAX_LEN_TOO_SMALL;   /* :14-17 */
    if (rc) { free(it); return rc; }
    uint32_t *sorted = (uint32_t *)malloc(m * sizeof(uint32_t));
    uint8_t *lens = (uint8_t *)malloc(m);
    for (size_t i = 0; i < m; i++) sorted[i] = it[i].key;
    pm_sorted(sorted, m, max_len, lens);
    memset(code_lens, 0, n);
    for (size_t i = 0; i < m; i++) code_lens[it[i].sym] = lens[i]; /* :21-27 */
    free(it); free(sorted); free(lens);
    return W3O_PM_OK;
}

void w3o_canonical(const uint8_t *code_lens, size_t n, uint16_t *codes_out, uint8_t *lens_out) { /* :87-117 */
    pm_item *it = (pm_item *)malloc((n ? n : 1) * sizeof(pm_item));
    size_t m = 0;
    unsigned max_len = 0;
    for (size_t i = 0; i < n; i++) {
        if (code_lens[i] > max_len) max_len = code_lens[i];
        if (code_lens[i] != 0) { it[m].sym = (uint32_t)i; it[m].key = code_lens[i]; m++; }
    }
    pm_stable_sort(it, m);                                        /* :92 sort_unstable_by on the lengths */
    uint16_t count_lens[260], codes[260];
    memset(count_lens, 0, sizeof count_lens); memset(codes, 0, sizeof codes);
    for (size_t i = 0; i < m; i++) count_lens[it[i].key] += 1;
    for (unsigned i = 0; i < max_len; i++) codes[i + 1] = (uint16_t)((codes[i] + count_lens[i]) << 1);   /* :107-109 */
    for (size_t i = 0; i < n; i++) { codes_out[i] = 0; lens_out[i] = 0; }
    for (size_t i = 0; i < m; i++) {                              /* :112-115 */
        const unsigned l = it[i].key;
        codes_out[it[i].sym] = codes[l]; lens_out[it[i].sym] = (uint8_t)l;
        codes[l] = (uint16_t)(codes[l] + 1);
    }
    free(it);
}

static uint16_t rev16(uint16_t v) { uint16_t r = 0; for (int i = 0; i < 16; i++) r = (uint16_t)((r << 1) | ((v >> i) & 1)); return r; }

int w3o_huff_tables_new(const uint8_t *buf, size_t n, uint8_t huff_size, uint8_t rem_huff_size, w3o_huff_tables *out) { /* huff_history.rs:17-55 */
    uint32_t counts[256], rem_counts[256];
    uint8_t lens[256];
    memset(counts, 0, sizeof counts); memset(rem_counts, 0, sizeof rem_counts);
    for (size_t i = 0; i < n; i++) counts[buf[i]] += 1;           /* helpers.rs:30-36 */
    int rc = w3o_package_merge(counts, 256, huff_size, lens);
    if (rc) return rc;
    w3o_canonical(lens, 256, out->code, out->len);
    for (int i = 0; i < 256; i++)                                  /* :21-25: reverse_bits().overflowing_shr(16 - len): amount taken mod 16 */
        out->code[i] = (uint16_t)(rev16(out->code[i]) >> ((16u - out->len[i]) & 15u));
    for (int byte = 0; byte < 256; byte++)                         /* :27-34 */
        for (int bit_len = 0; bit_len < 8; bit_len++) rem_counts[(1 << bit_len) | (byte >> (8 - bit_len))] += counts[byte];
    rc = w3o_package_merge(rem_counts, 256, rem_huff_size, lens);
    if (rc) return rc;
    w3o_canonical(lens, 256, out->rem_code, out->rem_len);
    for (int i = 0; i < 256; i++)
        out->rem_code[i] = (uint16_t)(rev16(out->rem_code[i]) >> ((16u - out->rem_len[i]) & 15u));
    return W3O_PM_OK;
}

/* ================================================================== */
/* History  history/raw_history.rs, history/ac_history.rs              */
/* ================================================================== */

void w3o_history_raw(w3o_history *h) { memset(h, 0, sizeof *h); h->kind = W3O_HIST_RAW; }
void w3o_history_ac(w3o_history *h, uint8_t max_bits, const w3o_stationary *m) {
    memset(h, 0, sizeof *h);
    h->kind = W3O_HIST_AC;
    h->max_bits = max_bits;
    h->model = *m;
}
void w3o_history_huff(w3o_history *h, const w3o_huff_tables *t) { /* huff_history.rs:44-50 */
    memset(h, 0, sizeof *h);
    h->kind = W3O_HIST_HUFF;
    h->huff = *t;
}
void w3o_history_update(w3o_history *h, uint8_t bit) {
    if (h->kind == W3O_HIST_RAW) { /* raw_history.rs:14-16 */
        h->raw_bits = (h->raw_bits << 1) | bit;
    } else { /* ac_history.rs:23-26, huff_history.rs:59-62 */
        h->bits = (h->bits << 1) | bit;
        h->pos += 1;
    }
}
uint32_t w3o_history_hash(w3o_history *h) {
    if (h->kind == W3O_HIST_RAW) return h->raw_bits; /* raw_history.rs:18-20 */
    if (h->kind == W3O_HIST_HUFF) { /* huff_history.rs:64-76 */
        const unsigned alignment = (unsigned)(h->pos & 7);
        if (alignment == 0) {   /* a byte has just been completed: append its code (u32: old bits fall off the top) */
            const uint8_t byte = (uint8_t)(h->bits & 255);
            const unsigned len = h->huff.len[byte];
            h->compressed_bits = (len >= 32 ? 0u : (h->compressed_bits << len)) | h->huff.code[byte];
        }
        const uint32_t mask = (1u << alignment) - 1u;
        const uint8_t rem_sym = (uint8_t)((h->bits & mask) | (1u << alignment));
        const unsigned len = h->huff.rem_len[rem_sym];
        return (len >= 32 ? 0u : (h->compressed_bits << len)) | h->huff.rem_code[rem_sym];
    }
    /* ac_history.rs:28-46 */
    w3o_ac ac; w3o_sink w;
    w3o_ac_new_coder(&ac);
    w3o_sink_init_entropy(&w, h->max_bits);
    stationary_align(&h->model, (uint8_t)(h->pos & 7));
    for (int i = 0; i < 64; i++) {
        uint8_t bit = (uint8_t)((h->bits >> i) & 1);
        if (w3o_ac_encode(&ac, bit, stationary_predict(&h->model), &w)) break;
    }
    /* state >> (32 - idx): idx==0 is a shift by 32 -> release-mode wrap gives state (=0) */
    unsigned sh = 32u - w.eidx;
    return sh >= 32 ? w.state : (w.state >> sh);
}

/* ================================================================== */
/* ACHistoryCached  history/ac_history_cached.rs:9-76                  */
/* The reference's memoised form of ACHistory, restated WITH its memo  */
/* (the product treats it as an alias of ACHistory: this is what the   */
/* equality test compares).  Rust's std HashMap<(u64, u8, u8), (EntropyWriter, ArithmeticCoder)> */
/* is an open-addressing table here; insert overwrites like :68, :71.  */
/* Integer behaviour is the RELEASE profile's (Cargo.toml:20-24: no    */
/* overflow checks): cache_size / 2 == 0 makes `c2 - 1` wrap to 255,   */
/* which no i in 0..64 equals (:67) — a debug build would panic there. */
/* ================================================================== */

typedef struct achc_entry {
    uint64_t bits; uint8_t alignment, level, used;        /* key (:15) */
    uint32_t state; uint8_t max_bits, idx; uint16_t rev_bits;   /* EntropyWriter (:79-85) */
    uint32_t x1, x2;                                       /* ArithmeticCoder (arithmetic_coder.rs:13-18; x is the decoder's) */
} achc_entry;

struct w3o_achc {
    uint64_t pos, bits;          /* :10-11 */
    uint8_t max_bits;            /* :12 */
    w3o_stationary model;        /* :13 */
    uint8_t cache_size;          /* :14 */
    achc_entry *tab; size_t cap, len;   /* :15 */
    uint64_t hits[2], misses;    /* (instrumentation for the tests: level-0 / level-1 hits, full runs) */
};

static size_t achc_slot(const w3o_achc *h, uint64_t bits, uint8_t alignment, uint8_t level) {
    uint64_t k = bits * 0x9E3779B97F4A7C15ull + ((uint64_t)alignment << 8 | level) * 0xC2B2AE3D27D4EB4Full;
    k ^= k >> 29;
    size_t i = (size_t)k & (h->cap - 1);
    while (h->tab[i].used && !(h->tab[i].bits == bits && h->tab[i].alignment == alignment && h->tab[i].level == level))
        i = (i + 1) & (h->cap - 1);
    return i;
}

static void achc_insert(w3o_achc *h, uint64_t bits, uint8_t alignment, uint8_t level, const w3o_sink *w, const w3o_ac *ac) {
    if ((h->len + 1) * 2 > h->cap) {   /* grow */
        achc_entry *old = h->tab; const size_t ocap = h->cap;
        h->cap = ocap ? ocap * 2 : 1024;
        h->tab = (achc_entry *)calloc(h->cap, sizeof(achc_entry));
        for (size_t i = 0; i < ocap; i++)
            if (old[i].used) h->tab[achc_slot(h, old[i].bits, old[i].alignment, old[i].level)] = old[i];
        free(old);
    }
    achc_entry *e = &h->tab[achc_slot(h, bits, alignment, level)];
    if (!e->used) h->len++;
    e->used = 1; e->bits = bits; e->alignment = alignment; e->level = level;
    e->state = w->state; e->max_bits = w->max_bits; e->idx = w->eidx; e->rev_bits = w->erev;
    e->x1 = ac->x1; e->x2 = ac->x2;
}

w3o_achc *w3o_achc_new(uint8_t max_bits, const w3o_stationary *m, uint8_t cache_size) {   /* :19-28 */
    w3o_achc *h = (w3o_achc *)calloc(1, sizeof *h);
    h->max_bits = max_bits; h->model = *m; h->cache_size = cache_size;
    return h;
}
void w3o_achc_free(w3o_achc *h) { if (h) { free(h->tab); free(h); } }
void w3o_achc_update(w3o_achc *h, uint8_t bit) {   /* :32-35 */
    h->bits = (h->bits << 1) | bit;
    h->pos += 1;
}
void w3o_achc_counts(const w3o_achc *h, uint64_t out[4]) { out[0] = h->hits[0]; out[1] = h->hits[1]; out[2] = h->misses; out[3] = h->len; }

uint32_t w3o_achc_hash(w3o_achc *h) {   /* :37-76 */
    uint8_t alignment = (uint8_t)(h->pos & 7);                                        /* :38 */
    const uint8_t c1 = h->cache_size, c2 = (uint8_t)(h->cache_size / 2);              /* :40 */
    const uint64_t m1 = (c1 >= 64 ? 0 : (1ull << c1)) - 1, m2 = (1ull << c2) - 1;     /* :41 (1u64 << 64 would panic: cache sizes stay below) */
    const uint64_t k1 = h->bits & m1, k2 = h->bits & m2;                              /* :42-45: keys (bits & m, alignment, level) */
    uint8_t start;
    w3o_sink writer; w3o_ac ac;
    const achc_entry *e = h->cap ? &h->tab[achc_slot(h, k1, alignment, 0)] : NULL;    /* :46 cache.get(&k1) */
    int level = 0;
    if (!e || !e->used) { e = h->cap ? &h->tab[achc_slot(h, k2, alignment, 1)] : NULL; level = 1; }   /* :48 cache.get(&k2) */
    if (e && e->used) {   /* :47 / :49: clones of the memoised writer and coder */
        start = level == 0 ? c1 : c2;
        w3o_sink_init_entropy(&writer, e->max_bits);
        writer.state = e->state; writer.eidx = e->idx; writer.erev = e->rev_bits;
        ac.x1 = e->x1; ac.x2 = e->x2; ac.x = 0;
        h->hits[level]++;
    } else {              /* :50-54 */
        start = 0;
        w3o_sink_init_entropy(&writer, h->max_bits);
        w3o_ac_new_coder(&ac);
        h->misses++;
    }
    alignment = (uint8_t)(((uint8_t)(alignment + 32) - start) & 7);                   /* :58 (u8 arithmetic) */
    stationary_align(&h->model, alignment);                                           /* :59 */
    for (uint8_t i = start; i < 64; i++) {                                            /* :60 */
        const uint8_t bit = (uint8_t)((h->bits >> i) & 1);                            /* :61 */
        if (w3o_ac_encode(&ac, bit, stationary_predict(&h->model), &writer)) break;   /* :62-65 */
        if (i == (uint8_t)(c2 - 1)) achc_insert(h, k2, (uint8_t)(h->pos & 7), 1, &writer, &ac);   /* :66-68 */
        if (i == (uint8_t)(c1 - 1)) achc_insert(h, k1, (uint8_t)(h->pos & 7), 0, &writer, &ac);   /* :69-71 */
    }
    const unsigned sh = 32u - writer.eidx;                                            /* :75; idx == 0: see w3o_history_hash */
    return sh >= 32 ? writer.state : (writer.state >> sh);
}

/* ================================================================== */
/* Models                                                              */
/* ================================================================== */

enum { M_ORDER0, M_ORDER1, M_ORDERN, M_ORDERN_ENTROPY, M_FROZEN, M_BEST2, M_SLOT, M_APM };

struct w3o_model {
    int kind;
    /* adaptive leaves */
    w3o_counter *stats; size_t nstats;
    uint32_t ctx, history; uint8_t alignment, bits, align_bits;
    w3o_history hist, hist0;
    /* sparse reset support for big tables (implementation detail: which
       counters left (0,0); does not change any output) */
    uint32_t *dirty; size_t ndirty, dirty_cap;
    /* composites */
    w3o_model *a, *b;
    /* M_SLOT (build-defined): byte history, current cell/slot, position inside the nibble */
    uint8_t order, log_cells, slot_id, bit_id, nib_ctx;
    uint64_t hist_bytes; uint32_t c0;
    w3o_cell *cells, *cur;
    /* M_APM (build-defined) */
    uint16_t *apm_t; uint32_t apm_rows; uint8_t apm_ctx, apm_rate; uint32_t apm_c0, apm_c1;
};

#define DIRTY_THRESHOLD ((size_t)1 << 16)

static w3o_model *leaf_new(int kind, uint8_t bits, uint8_t align) {
    w3o_model *m = (w3o_model *)calloc(1, sizeof *m);
    m->kind = kind; m->bits = bits; m->align_bits = align;
    m->nstats = (size_t)1 << bits;
    m->stats = (w3o_counter *)calloc(m->nstats, sizeof(w3o_counter));
    if (m->nstats > DIRTY_THRESHOLD) {
        m->dirty_cap = 1 << 16;
        m->dirty = (uint32_t *)malloc(m->dirty_cap * sizeof(uint32_t));
    }
    return m;
}
w3o_model *w3o_order0(void) { return leaf_new(M_ORDER0, 11, 3); }            /* order0.rs:11-18 */
w3o_model *w3o_order1(void) { return leaf_new(M_ORDER1, 19, 3); }            /* order1.rs:12-19 */
w3o_model *w3o_ordern(uint8_t bits, uint8_t align) { return leaf_new(M_ORDERN, bits, align); } /* ordern.rs:14-23 */
w3o_model *w3o_ordern_entropy(uint8_t bits, uint8_t align, const w3o_history *h) { /* ordern_entropy.rs:15-24 */
    w3o_model *m = leaf_new(M_ORDERN_ENTROPY, bits, align);
    m->hist = *h; m->hist0 = *h;
    return m;
}
w3o_model *w3o_frozen(w3o_model *adaptive) { /* frozen.rs:7-11 */
    w3o_model *m = (w3o_model *)calloc(1, sizeof *m);
    m->kind = M_FROZEN; m->a = adaptive;
    return m;
}
w3o_model *w3o_best_of_two(w3o_model *m1, w3o_model *m2) { /* mod.rs:57-60 */
    w3o_model *m = (w3o_model *)calloc(1, sizeof *m);
    m->kind = M_BEST2; m->a = m1; m->b = m2;
    return m;
}
static void slot_reset(w3o_model *m);
static void slot_update(w3o_model *m, uint8_t bit);
static void apm_reset(w3o_model *m);
static void apm_update(w3o_model *m, uint8_t bit);
static uint16_t apm_pp(const w3o_model *m, uint16_t p, uint32_t *idx);

w3o_model *w3o_model_clone_fresh(const w3o_model *m) {
    switch (m->kind) {
    case M_FROZEN: return w3o_frozen(w3o_model_clone_fresh(m->a));
    case M_BEST2:  return w3o_best_of_two(w3o_model_clone_fresh(m->a), w3o_model_clone_fresh(m->b));
    case M_ORDERN_ENTROPY: return w3o_ordern_entropy(m->bits, m->align_bits, &m->hist0);
    case M_SLOT: return w3o_slot_model(m->order, m->log_cells);
    case M_APM:  return w3o_apm(w3o_model_clone_fresh(m->a), m->apm_ctx, m->apm_rate);
    default: return leaf_new(m->kind, m->bits, m->align_bits);
    }
}
void w3o_model_reset(w3o_model *m) {
    if (m->kind == M_FROZEN) { w3o_model_reset(m->a); return; }
    if (m->kind == M_BEST2) { w3o_model_reset(m->a); w3o_model_reset(m->b); return; }
    if (m->kind == M_SLOT) { slot_reset(m); return; }
    if (m->kind == M_APM) { w3o_model_reset(m->a); apm_reset(m); return; }
    if (m->dirty) {
        for (size_t i = 0; i < m->ndirty; i++) { m->stats[m->dirty[i]].data[0] = 0; m->stats[m->dirty[i]].data[1] = 0; }
        m->ndirty = 0;
    } else {
        memset(m->stats, 0, m->nstats * sizeof(w3o_counter));
    }
    m->ctx = 0; m->history = 0; m->alignment = 0;
    m->hist = m->hist0;
}
void w3o_model_free(w3o_model *m) {
    if (!m) return;
    w3o_model_free(m->a); w3o_model_free(m->b);
    free(m->stats); free(m->dirty); free(m->cells); free(m->apm_t); free(m);
}

uint16_t w3o_opinion_mix(uint16_t p1, uint16_t p2) { /* opinion_mixer2.rs:5-10 */
    const uint16_t HALF = 1u << 15;
    uint16_t d1 = p1 >= HALF ? p1 - HALF : HALF - p1;
    uint16_t d2 = p2 >= HALF ? p2 - HALF : HALF - p2;
    return d1 >= d2 ? p1 : p2;
}

uint16_t w3o_model_predict(const w3o_model *m) {
    switch (m->kind) {
    case M_FROZEN: return w3o_model_predict(m->a);                          /* frozen.rs:14-16 */
    case M_BEST2:  return w3o_opinion_mix(w3o_model_predict(m->a), w3o_model_predict(m->b)); /* mod.rs:67-69 */
    case M_SLOT:   return w3o_st_p(w3o_slot_get_state(m->cur, m->slot_id, m->bit_id, m->nib_ctx));
    case M_APM:    { uint32_t idx; return apm_pp(m, w3o_model_predict(m->a), &idx); }
    default:       return w3o_counter_p(&m->stats[m->ctx]);                 /* order0.rs:22-24 etc. */
    }
}

static void leaf_adapt(w3o_model *m, uint8_t bit) { /* order0.rs:26-28, ordern.rs:31-33 ... */
    w3o_counter *c = &m->stats[m->ctx];
    if (m->dirty && c->data[0] == 0 && c->data[1] == 0) {
        if (m->ndirty == m->dirty_cap) {
            m->dirty_cap *= 2;
            m->dirty = (uint32_t *)realloc(m->dirty, m->dirty_cap * sizeof(uint32_t));
        }
        m->dirty[m->ndirty++] = m->ctx;
    }
    w3o_counter_update(c, bit);
}

static void leaf_advance(w3o_model *m, uint8_t bit) { /* AdaptiveModel::update */
    switch (m->kind) {
    case M_ORDER0: { /* order0.rs:30-34: u8 history, alignment<<8 | history */
        uint8_t h = (uint8_t)(((uint8_t)m->history << 1) | bit);
        m->history = h;
        m->alignment = (uint8_t)((m->alignment + 1) % 8);
        m->ctx = ((uint32_t)m->alignment << 8) | h;
        break;
    }
    case M_ORDER1: { /* order1.rs:31-35: u16 history, alignment<<16 | history */
        uint16_t h = (uint16_t)(((uint16_t)m->history << 1) | bit);
        m->history = h;
        m->alignment = (uint8_t)((m->alignment + 1) % 8);
        m->ctx = ((uint32_t)m->alignment << 16) | h;
        break;
    }
    case M_ORDERN: { /* ordern.rs:35-43 */
        uint32_t mask_bits = (uint32_t)m->bits - m->align_bits;
        uint32_t mask = (uint32_t)(((uint64_t)1 << mask_bits) - 1);
        uint8_t amask = (uint8_t)((1u << m->align_bits) - 1);
        m->history = ((m->history << 1) | bit) & mask;
        m->alignment = (uint8_t)((m->alignment + 1) & amask);
        m->ctx = (m->history << m->align_bits) | m->alignment;
        break;
    }
    default: { /* ordern_entropy.rs:36-45 */
        uint32_t mask_bits = (uint32_t)m->bits - m->align_bits;
        uint32_t mask = (uint32_t)(((uint64_t)1 << mask_bits) - 1);
        uint8_t amask = (uint8_t)((1u << m->align_bits) - 1);
        w3o_history_update(&m->hist, bit);
        m->alignment = (uint8_t)((m->alignment + 1) & amask);
        uint32_t hash = w3o_history_hash(&m->hist) & mask;
        m->ctx = (hash << m->align_bits) | m->alignment;
        break;
    }
    }
}

void w3o_model_update(w3o_model *m, uint8_t bit) {
    switch (m->kind) {
    case M_FROZEN: leaf_advance(m->a, bit); break;                               /* frozen.rs:18-20 */
    case M_BEST2:  w3o_model_update(m->a, bit); w3o_model_update(m->b, bit); break; /* mod.rs:71-74 */
    case M_SLOT:   slot_update(m, bit); break;
    case M_APM:    apm_update(m, bit); break;
    default:       leaf_adapt(m, bit); leaf_advance(m, bit); break;              /* mod.rs:28-31 */
    }
}

/* ================================================================== */
/* Bit loops  main.rs:103-111, 131-140                                 */
/* ================================================================== */

static void encode_into(w3o_model *m, const uint8_t *in, size_t n, w3o_sink *w) {
    w3o_ac ac;
    w3o_ac_new_coder(&ac);
    for (size_t k = 0; k < n; k++) {
        uint8_t byte = in[k];
        for (int s = 7; s >= 0; s--) {
            uint8_t bit = (byte >> s) & 1;
            uint16_t p = w3o_model_predict(m);
            w3o_model_update(m, bit);
            w3o_ac_encode(&ac, bit, p, w);
        }
    }
    w3o_ac_flush(&ac, w);
}

uint8_t *w3o_encode_stream(w3o_model *m, const uint8_t *in, size_t n, size_t *out_len) {
    w3o_sink w;
    w3o_sink_init_bytes(&w);
    encode_into(m, in, n, &w);
    *out_len = w.len;
    return w.buf; /* caller frees */
}

uint64_t w3o_encode_stats(w3o_model *m, const uint8_t *in, size_t n) {
    w3o_sink w;
    w3o_sink_init_stats(&w);
    encode_into(m, in, n, &w);
    return w3o_stats_result(&w);
}
uint64_t w3o_encode_stats_bits(w3o_model *m, const uint8_t *in, size_t n) { /* ACStats::bit_count itself (helpers.rs:62, :79-82) */
    w3o_sink w;
    w3o_sink_init_stats(&w);
    encode_into(m, in, n, &w);
    return w.bit_count;
}

void w3o_predict_all(w3o_model *m, const uint8_t *in, size_t n, uint16_t *p_out) {
    for (size_t k = 0; k < n; k++)
        for (int s = 7; s >= 0; s--) {
            *p_out++ = w3o_model_predict(m);
            w3o_model_update(m, (in[k] >> s) & 1);
        }
}

void w3o_decode_stream(w3o_model *m, const uint8_t *in, size_t in_len, uint8_t *out, size_t n) {
    w3o_reader r; w3o_ac ac;
    w3o_reader_init(&r, in, in_len);
    w3o_ac_new_decoder(&ac, &r);
    for (size_t k = 0; k < n; k++) {
        uint8_t byte = 0;
        for (int s = 0; s < 8; s++) {
            uint16_t p = w3o_model_predict(m);
            uint8_t bit = w3o_ac_decode(&ac, p, &r);
            w3o_model_update(m, bit);
            byte = (uint8_t)((byte << 1) | bit);
        }
        out[k] = byte;
    }
}

/* main.rs:14-15,95-96: b"w30i" + u64 BE length + stream */
uint8_t *w3o_compress_container(w3o_model *m, const uint8_t *in, size_t n, size_t *out_len) {
    w3o_sink w;
    w3o_sink_init_bytes(&w);
    static const uint8_t magic[4] = {'w', '3', '0', 'i'};
    for (int i = 0; i < 4; i++) sink_push_byte(&w, magic[i]);
    for (int i = 7; i >= 0; i--) sink_push_byte(&w, (uint8_t)((uint64_t)n >> (8 * i)));
    encode_into(m, in, n, &w);
    *out_len = w.len;
    return w.buf;
}

int w3o_decompress_container(w3o_model *m, const uint8_t *in, size_t in_len, uint8_t **out, size_t *out_len) {
    if (in_len < 12) return -1;                                  /* read_exact fails, main.rs:121 */
    if (memcmp(in, "w30i", 4) != 0) return -2;                   /* assert_eq! magic, main.rs:123-124 */
    uint64_t len = 0;
    for (int i = 0; i < 8; i++) len = (len << 8) | in[4 + i];
    uint8_t *o = (uint8_t *)malloc(len ? len : 1);
    w3o_decode_stream(m, in + 12, in_len - 12, o, (size_t)len);
    *out = o; *out_len = (size_t)len;
    return 0;
}

/* ================================================================== */
/* Block mode (build-defined container element)      */
/* ================================================================== */

typedef struct {
    const w3o_model *proto; const uint8_t *in; size_t n, bs, nblocks;
    uint8_t **streams; size_t *lens;
    /* decode */
    const uint8_t *cin; const size_t *offs; const uint32_t *blens; uint8_t *out; uint64_t orig_len;
    int decode;
    size_t next; pthread_mutex_t mu;
} blk_job;

static void *blk_worker(void *arg) {
    blk_job *j = (blk_job *)arg;
    w3o_model *m = w3o_model_clone_fresh(j->proto);
    for (;;) {
        pthread_mutex_lock(&j->mu);
        size_t b = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (b >= j->nblocks) break;
        size_t off = b * j->bs;
        w3o_model_reset(m);
        if (!j->decode) {
            size_t len = j->n - off < j->bs ? j->n - off : j->bs;
            j->streams[b] = w3o_encode_stream(m, j->in + off, len, &j->lens[b]);
        } else {
            size_t len = (size_t)j->orig_len - off < j->bs ? (size_t)j->orig_len - off : j->bs;
            w3o_decode_stream(m, j->cin + j->offs[b], j->blens[b], j->out + off, len);
        }
    }
    w3o_model_free(m);
    return NULL;
}

static void run_workers(blk_job *j, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    pthread_mutex_init(&j->mu, NULL);
    if (nthreads == 1) { blk_worker(j); return; }
    pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int i = 0; i < nthreads; i++) pthread_create(&t[i], NULL, blk_worker, j);
    for (int i = 0; i < nthreads; i++) pthread_join(t[i], NULL);
    free(t);
}

int w3o_encode_blocks(const w3o_model *proto, const uint8_t *in, size_t n, size_t block_size,
                      uint8_t *out, size_t out_cap, size_t *out_len, uint32_t *block_lens, int nthreads) {
    size_t nb = block_size ? (n + block_size - 1) / block_size : 0;
    blk_job j; memset(&j, 0, sizeof j);
    j.proto = proto; j.in = in; j.n = n; j.bs = block_size; j.nblocks = nb;
    j.streams = (uint8_t **)calloc(nb ? nb : 1, sizeof(uint8_t *));
    j.lens = (size_t *)calloc(nb ? nb : 1, sizeof(size_t));
    run_workers(&j, nthreads);
    size_t total = 0; int rc = 0;
    for (size_t b = 0; b < nb; b++) {
        if (total + j.lens[b] <= out_cap) memcpy(out + total, j.streams[b], j.lens[b]); else rc = -1;
        block_lens[b] = (uint32_t)j.lens[b];
        total += j.lens[b];
        free(j.streams[b]);
    }
    *out_len = total;
    free(j.streams); free(j.lens);
    return rc;
}

int w3o_decode_blocks(const w3o_model *proto, const uint8_t *in, const uint32_t *block_lens,
                      size_t nblocks, size_t block_size, uint64_t orig_len, uint8_t *out, int nthreads) {
    blk_job j; memset(&j, 0, sizeof j);
    size_t *offs = (size_t *)calloc(nblocks ? nblocks : 1, sizeof(size_t));
    size_t acc = 0;
    for (size_t b = 0; b < nblocks; b++) { offs[b] = acc; acc += block_lens[b]; }
    j.proto = proto; j.bs = block_size; j.nblocks = nblocks; j.decode = 1;
    j.cin = in; j.offs = offs; j.blens = block_lens; j.out = out; j.orig_len = orig_len;
    run_workers(&j, nthreads);
    free(offs);
    return 0;
}

/* ================================================================== */
/* NaiveStateTable  state_table/naive.rs                               */
/* ================================================================== */

#define MAX_LEVEL 44
static w3o_state_entry g_st[W3O_ST_SIZE];
static w3o_state_entry g_aux[W3O_ST_AUX];
static pthread_once_t g_st_once = PTHREAD_ONCE_INIT;

static uint16_t st_calc_prob(uint16_t count, uint16_t total) { /* naive.rs:107-113 (floored) */
    uint64_t p = ((uint64_t)1 << 16) * ((uint64_t)count + 1) / ((uint64_t)total + 2);
    return (uint16_t)p;
}

static void st_next_nodes(size_t level, size_t filled, size_t node, uint16_t next[2]) { /* naive.rs:79-104 */
    if (level == MAX_LEVEL) {
        size_t next_level = ((level + 2) / 2) - 1; /* 22 */
        size_t next_idx = ((node + 2) / 2) - 1;
        uint16_t cur = (uint16_t)(filled + node);
        uint16_t nn = (uint16_t)(next_idx + (next_level - 1) * next_level / 2);
        next[0] = nn; next[1] = nn;
        if (node == 0) next[0] = cur;
        if (node == level - 1) next[1] = cur;
        return;
    }
    uint16_t nn = (uint16_t)(filled + level + node);
    next[0] = nn; next[1] = (uint16_t)(nn + 1);
}

static void st_build(void) {
    /* gen_auxiliary_table  naive.rs:52-77 */
    size_t filled = 0;
    for (size_t level = 1; level <= MAX_LEVEL; level++) {
        for (size_t node = 0; node < level; node++) {
            g_aux[filled + node].prob = st_calc_prob((uint16_t)node, (uint16_t)(level - 1));
            st_next_nodes(level, filled, node, g_aux[filled + node].next);
        }
        filled += level;
    }
    /* gen_table  naive.rs:17-50 */
    const uint16_t OFFSET = W3O_ST_AUX, HALF = 1u << 15;
    uint16_t a = 3, b = a + OFFSET, c = a + 2 * OFFSET, d = a + 3 * OFFSET;
    g_st[0].prob = HALF; g_st[0].next[0] = 1; g_st[0].next[1] = 2;
    g_st[1].prob = HALF; g_st[1].next[0] = a; g_st[1].next[1] = b;
    g_st[2].prob = HALF; g_st[2].next[0] = c; g_st[2].next[1] = d;
    for (size_t i = 0; i < W3O_ST_AUX; i++) {
        uint16_t n0 = g_aux[i].next[0], n1 = g_aux[i].next[1], p = g_aux[i].prob;
        g_st[a + i].prob = p; g_st[a + i].next[0] = a + n0; g_st[a + i].next[1] = b + n1;
        g_st[b + i].prob = p; g_st[b + i].next[0] = c + n0; g_st[b + i].next[1] = d + n1;
        g_st[c + i].prob = p; g_st[c + i].next[0] = a + n0; g_st[c + i].next[1] = b + n1;
        g_st[d + i].prob = p; g_st[d + i].next[0] = c + n0; g_st[d + i].next[1] = d + n1;
    }
}

const w3o_state_entry *w3o_state_table(void) { pthread_once(&g_st_once, st_build); return g_st; }
const w3o_state_entry *w3o_state_table_aux(void) { pthread_once(&g_st_once, st_build); return g_aux; }
uint16_t w3o_st_next(uint16_t s, uint8_t bit) { return w3o_state_table()[s].next[bit]; } /* mod.rs:43-45 */
uint16_t w3o_st_p(uint16_t s) { return w3o_state_table()[s].prob; }                      /* mod.rs:47-49 */
void w3o_st_next4(const uint16_t s[4], uint8_t nib, uint16_t out[4]) { /* mod.rs:5-12 */
    out[0] = w3o_st_next(s[0], nib >> 3);
    out[1] = w3o_st_next(s[1], (nib >> 2) & 1);
    out[2] = w3o_st_next(s[2], (nib >> 1) & 1);
    out[3] = w3o_st_next(s[3], nib & 1);
}
void w3o_st_p4(const uint16_t s[4], uint16_t out[4]) { /* mod.rs:15-22 */
    for (int i = 0; i < 4; i++) out[i] = w3o_st_p(s[i]);
}

/* ================================================================== */
/* HashMap / Cell / Slot  hashmap.rs                                   */
/* ================================================================== */

uint32_t w3o_hashmap_log_cell_count(size_t size_bytes) { /* hashmap.rs:8-9 (f64 log2, truncated) */
    double v = log2((double)size_bytes) - log2((double)sizeof(w3o_cell));
    return (uint32_t)v;
}
uint64_t w3o_hashmap_cell_index(uint64_t hash, uint32_t log_cell_count) { /* hashmap.rs:26 (high bits) */
    return hash >> (64 - log_cell_count);
}
uint8_t w3o_cell_get_slot(const w3o_cell *c, uint64_t hash) { /* hashmap.rs:42-71 */
    uint64_t hc = 0;
    for (int i = 0; i < 6; i++) hc = (hc << 8) | c->hashes[i];
    uint64_t mask = (1u << 12) - 1, h = hash & mask;
    if (h == (hc & mask)) return 3;
    if (h == ((hc >> 12) & mask)) return 2;
    if (h == ((hc >> 24) & mask)) return 1;
    if (h == ((hc >> 36) & mask)) return 0;
    return 1; /* TODO in the reference: miss -> slot 1, tag not stored (hashmap.rs:64-68) */
}
void w3o_slot_get_idx(uint8_t id, uint8_t bit_id, uint8_t nib_ctx, uint32_t *abs_idx, int *parity) { /* :80-84 */
    uint32_t idx = (3u << bit_id) + 3u * nib_ctx + 45u * id - 3u;
    *parity = (idx & 1) == 1;
    *abs_idx = idx >> 1;
}
uint16_t w3o_slot_get_state(const w3o_cell *c, uint8_t id, uint8_t bit_id, uint8_t nib_ctx) { /* :86-97 */
    uint32_t a; int par;
    w3o_slot_get_idx(id, bit_id, nib_ctx, &a, &par);
    uint16_t st = (uint16_t)((c->slots[a] << 8) | c->slots[a + 1]);
    return par ? (st & 0xFFF) : (st >> 4);
}
void w3o_slot_set_state(w3o_cell *c, uint8_t id, uint8_t bit_id, uint8_t nib_ctx, uint16_t ns) { /* :99-112 */
    uint32_t a; int par;
    w3o_slot_get_idx(id, bit_id, nib_ctx, &a, &par);
    if (!par) {
        uint16_t v = (uint16_t)(ns << 4);
        c->slots[a] = (uint8_t)(v >> 8);
        c->slots[a + 1] = (uint8_t)((v & 0xFF) | (c->slots[a + 1] & 15));
    } else {
        c->slots[a] = (uint8_t)((ns >> 8) | (c->slots[a] & (15 << 4)));
        c->slots[a + 1] = (uint8_t)(ns & 0xFF);
    }
}
void w3o_slot_set_nib(w3o_cell *c, uint8_t id, uint8_t nib, const uint16_t st[4]) { /* :114-119 */
    w3o_slot_set_state(c, id, 0, 0, st[0]);
    w3o_slot_set_state(c, id, 1, nib >> 3, st[1]);
    w3o_slot_set_state(c, id, 2, nib >> 2, st[2]);
    w3o_slot_set_state(c, id, 3, nib >> 1, st[3]);
}
void w3o_slot_get_nib(const w3o_cell *c, uint8_t id, uint8_t nib, uint16_t out[4]) { /* :121-128 */
    out[0] = w3o_slot_get_state(c, id, 0, 0);
    out[1] = w3o_slot_get_state(c, id, 1, nib >> 3);
    out[2] = w3o_slot_get_state(c, id, 2, nib >> 2);
    out[3] = w3o_slot_get_state(c, id, 3, nib >> 1);
}

/* ================================================================== */
/* BUILD-DEFINED models (SURVEY §8 A19 ii-v).  The reference has the    */
/* primitives (state table, Cell/Slot) but no model that uses them, no  */
/* replacement policy, no APM; these definitions are this build's own,  */
/* integer-only, and "parity unpinned" (DESIGN.md §2.4).                */
/* ================================================================== */

/* observation-count proxy of a state: 0 for the root, 1 for states 1/2, level+1 inside a lattice copy
 * (naive.rs:19-27,52-77: level L holds the nodes reached after L-1 lattice steps) */
static uint8_t g_conf[W3O_ST_SIZE];
static pthread_once_t g_conf_once = PTHREAD_ONCE_INIT;
static void conf_build(void) {
    g_conf[0] = 0; g_conf[1] = 1; g_conf[2] = 1;
    size_t filled = 0;
    for (size_t level = 1; level <= MAX_LEVEL; level++) {
        for (size_t node = 0; node < level; node++)
            for (int copy = 0; copy < 4; copy++) g_conf[3 + copy * W3O_ST_AUX + filled + node] = (uint8_t)(level + 1);
        filled += level;
    }
}
uint8_t w3o_st_conf(uint16_t state) { pthread_once(&g_conf_once, conf_build); return g_conf[state]; }

/* splitmix64 finaliser over (order, previous `order` bytes, nibble marker) */
uint64_t w3o_slot_hash(uint8_t order, uint64_t hist_bytes, int second, uint32_t hi_nib) {
    uint64_t k = order ? (hist_bytes & ((1ull << (8 * order)) - 1ull)) : 0;   /* order <= 7 */
    k = (k << 8) | (second ? (0x10u | hi_nib) : 0u);
    k += (uint64_t)(order + 1) * 0x9E3779B97F4A7C15ull;
    k ^= k >> 30; k *= 0xBF58476D1CE4E5B9ull;
    k ^= k >> 27; k *= 0x94D049BB133111EBull;
    k ^= k >> 31;
    return k;
}

static uint32_t cell_tag(const w3o_cell *c, int id) { /* hashmap.rs:43-63: id 3 = lowest 12 bits of the BE concat */
    uint64_t hc = 0;
    for (int i = 0; i < 6; i++) hc = (hc << 8) | c->hashes[i];
    return (uint32_t)((hc >> (12 * (3 - id))) & 0xFFF);
}
static void cell_set_tag(w3o_cell *c, int id, uint32_t tag) {
    uint64_t hc = 0;
    for (int i = 0; i < 6; i++) hc = (hc << 8) | c->hashes[i];
    int sh = 12 * (3 - id);
    hc = (hc & ~(0xFFFull << sh)) | ((uint64_t)tag << sh);
    for (int i = 5; i >= 0; i--) { c->hashes[i] = (uint8_t)hc; hc >>= 8; }
}

/* Locate the slot of a nibble context.  Hit: Cell::get_slot (hashmap.rs:42-63).  Miss: the policy the
 * reference leaves as "TODO: Select min and set new hash" (hashmap.rs:64-68) — the victim is the slot whose
 * first-bit state has the fewest observations (candidates in the order 1,0,2,3, so an empty cell gives the
 * PoC's slot 1, hashslots.md:24), its tag is set and its 15 states are cleared. */
static void slot_select(w3o_model *m, int second, uint32_t hi_nib) {
    uint64_t h = w3o_slot_hash(m->order, m->hist_bytes, second, hi_nib);
    w3o_cell *c = &m->cells[w3o_hashmap_cell_index(h, m->log_cells)];
    uint32_t tag = (uint32_t)(h & 0xFFF);
    int id = -1;
    for (int k = 3; k >= 0; k--) if (cell_tag(c, k) == tag) { id = k; break; }
    if (id < 0) {
        static const int cand[4] = {1, 0, 2, 3};
        int best = 256;
        for (int k = 0; k < 4; k++) {
            int conf = w3o_st_conf(w3o_slot_get_state(c, (uint8_t)cand[k], 0, 0));
            if (conf < best) { best = conf; id = cand[k]; }
        }
        cell_set_tag(c, id, tag);
        for (int bit_id = 0; bit_id < 4; bit_id++)
            for (int ctx = 0; ctx < (1 << bit_id); ctx++) w3o_slot_set_state(c, (uint8_t)id, (uint8_t)bit_id, (uint8_t)ctx, 0);
    }
    m->cur = c; m->slot_id = (uint8_t)id; m->bit_id = 0; m->nib_ctx = 0;
}

static void slot_reset(w3o_model *m) {
    memset(m->cells, 0, sizeof(w3o_cell) << m->log_cells);
    m->hist_bytes = 0; m->c0 = 1;
    slot_select(m, 0, 0);
}

w3o_model *w3o_slot_model(uint8_t order, uint8_t log_cells) {
    w3o_model *m = (w3o_model *)calloc(1, sizeof *m);
    m->kind = M_SLOT; m->order = order; m->log_cells = log_cells;
    m->cells = (w3o_cell *)malloc(sizeof(w3o_cell) << log_cells);
    slot_reset(m);
    return m;
}

static void slot_update(w3o_model *m, uint8_t bit) {
    uint16_t st = w3o_slot_get_state(m->cur, m->slot_id, m->bit_id, m->nib_ctx);
    w3o_slot_set_state(m->cur, m->slot_id, m->bit_id, m->nib_ctx, w3o_st_next(st, bit));   /* adapt */
    m->c0 = (m->c0 << 1) | bit;                                                             /* advance */
    m->nib_ctx = (uint8_t)((m->nib_ctx << 1) | bit);
    m->bit_id++;
    if (m->bit_id == 4) {
        if (m->c0 >= 256) { m->hist_bytes = (m->hist_bytes << 8) | (m->c0 & 0xFF); m->c0 = 1; slot_select(m, 0, 0); }
        else slot_select(m, 1, m->c0 & 15);
    }
}

/* ---- stretch / squash: integer-only logistic in units of 1/256 nat ------------------------------- */
static uint16_t g_squash[4095];   /* index d + 2047 */
static int16_t  g_stretch[4096];  /* index p >> 4    */
static pthread_once_t g_ss_once = PTHREAD_ONCE_INIT;
static void ss_build(void) {
    const uint64_t K = 0xFF007FD5ull;            /* round(2^32 * e^(-1/256)) */
    uint64_t e = 1ull << 32;                     /* e^(-d/256) in Q32 */
    for (int d = 0; d <= 2047; d++) {
        uint64_t den = (1ull << 32) + e;
        uint64_t q = ((1ull << 48) + den / 2) / den;
        if (q > 65535) q = 65535;
        g_squash[2047 + d] = (uint16_t)q;
        uint64_t lo = 65536 - q;
        g_squash[2047 - d] = (uint16_t)(lo < 1 ? 1 : lo);
        e = (e * K) >> 32;
    }
    g_squash[2047] = 32768;
    int d = -2047;
    for (int q = 0; q < 4096; q++) {             /* smallest d with squash(d) >= 16 q + 8 */
        uint32_t want = (uint32_t)q * 16 + 8;
        while (d < 2047 && g_squash[d + 2047] < want) d++;
        g_stretch[q] = (int16_t)d;
    }
}
uint16_t w3o_squash(int d) {
    pthread_once(&g_ss_once, ss_build);
    if (d < -2047) d = -2047;
    if (d > 2047) d = 2047;
    return g_squash[d + 2047];
}
int w3o_stretch(uint16_t p) { pthread_once(&g_ss_once, ss_build); return g_stretch[p >> 4]; }

/* ---- APM ------------------------------------------------------------------------------------------ */
static void apm_reset(w3o_model *m) {
    for (uint32_t r = 0; r < m->apm_rows; r++)
        for (int j = 0; j < 33; j++) m->apm_t[r * 33 + j] = w3o_squash((j - 16) * 128);
    m->apm_c0 = 1; m->apm_c1 = 0;
}
w3o_model *w3o_apm(w3o_model *input, uint8_t ctx_kind, uint8_t rate) {
    w3o_model *m = (w3o_model *)calloc(1, sizeof *m);
    m->kind = M_APM; m->a = input; m->apm_ctx = ctx_kind; m->apm_rate = rate;
    m->apm_rows = ctx_kind == W3O_APM_ORDER1 ? 65536u : 256u;
    m->apm_t = (uint16_t *)malloc((size_t)m->apm_rows * 33 * sizeof(uint16_t));
    apm_reset(m);
    return m;
}
static uint16_t apm_pp(const w3o_model *m, uint16_t p, uint32_t *idx) {
    uint32_t row = m->apm_ctx == W3O_APM_ORDER1 ? (m->apm_c0 | (m->apm_c1 << 8)) : m->apm_c0;
    uint32_t pos = (uint32_t)(w3o_stretch(p) + 2048) * 32u;
    uint32_t j = pos >> 12, w = pos & 4095u;
    const uint16_t *t = m->apm_t + row * 33u;
    uint32_t pa = ((uint32_t)t[j] * (4096u - w) + (uint32_t)t[j + 1] * w) >> 12;
    *idx = row * 33u + j + (w >> 11);
    uint32_t out = ((uint32_t)p + 3u * pa + 2u) >> 2;
    return (uint16_t)(out < 1 ? 1 : out > 65535 ? 65535 : out);
}
static void apm_update(w3o_model *m, uint8_t bit) {
    uint32_t idx;
    (void)apm_pp(m, w3o_model_predict(m->a), &idx);     /* the input's prediction for THIS step (before its update) */
    int32_t t = m->apm_t[idx], target = bit ? 65535 : 0, delta = target - t;
    int32_t step = delta >= 0 ? (delta >> m->apm_rate) : -((-delta + (1 << m->apm_rate) - 1) >> m->apm_rate);  /* floor */
    m->apm_t[idx] = (uint16_t)(t + step);
    w3o_model_update(m->a, bit);
    m->apm_c0 = (m->apm_c0 << 1) | bit;
    if (m->apm_c0 >= 256) { m->apm_c1 = m->apm_c0 & 0xFF; m->apm_c0 = 1; }
}
