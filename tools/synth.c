/* synth.c — seeded synthetic corpora for bench.py and the full-size tests
 * (no enwik/Silesia exists in this pipeline).
 *   w3s_text : "enwik-shaped" — Zipf-distributed pseudo-English words, wiki/XML
 *              markup tokens, digit runs, newlines every 40-120 chars.
 *   w3s_mixed: "Silesia-shaped" — segments of text, random bytes, noisy i32
 *              ramps, zero padding, repeated 4 KiB records.
 * Output depends only on (seed, absolute chunk index): chunks of 1 MiB are
 * generated independently, so any thread count gives identical bytes.
 * Bench/test utility: not part of the hot path, not part of the oracle. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CHUNK (1u << 20)
#define NWORDS 8192

static inline uint64_t rng_next(uint64_t *s) { /* splitmix64 */
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

typedef struct { char w[NWORDS][16]; uint8_t len[NWORDS]; uint32_t cum[NWORDS]; } vocab_t;
static vocab_t g_vocab; static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void vocab_build(void) {
    static const char *syl[] = {"th","e","an","in","er","on","re","ed","st","ar","ti","al","or","is","at","en","le","ou","ng","co",
        "de","ra","li","ic","ion","ent","pro","qu","mp","ly","a","o","i","s","t","es","ea","ur","ch","wh","ma","ne","se","ha","ve",
        "we","no","be","to","of","it","un","ab","ow","ir","ul","ck","sh","ph","gr"};
    const int nsyl = sizeof(syl) / sizeof(syl[0]);
    uint64_t s = 0x1234ABCDull;
    static const char *common[] = {"the","of","and","in","to","a","is","was","for","as","on","with","by","that","it","from","at","his","an","are"};
    double total = 0, wgt[NWORDS];
    for (int k = 0; k < NWORDS; k++) {
        char *w = g_vocab.w[k]; int L = 0;
        if (k < 20) { strcpy(w, common[k]); L = (int)strlen(w); }
        else {
            int ns = 1 + (int)(rng_next(&s) % 4) + (k > 2000);
            for (int j = 0; j < ns && L < 12; j++) { const char *y = syl[rng_next(&s) % nsyl]; int l = (int)strlen(y); memcpy(w + L, y, l); L += l; }
            if ((rng_next(&s) & 15) == 0) w[0] = (char)(w[0] - 32); /* capitalised */
            w[L] = 0;
        }
        g_vocab.len[k] = (uint8_t)L;
        double x = 1.0 / ((double)k + 2.7); wgt[k] = x; total += x;
    }
    double acc = 0;
    for (int k = 0; k < NWORDS; k++) { acc += wgt[k] / total; g_vocab.cum[k] = (uint32_t)(acc * 4294967295.0); }
    g_vocab.cum[NWORDS - 1] = 0xFFFFFFFFu;
}

static inline int pick_word(uint64_t *s) {
    uint32_t r = (uint32_t)rng_next(s);
    int lo = 0, hi = NWORDS - 1;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (g_vocab.cum[mid] < r) lo = mid + 1; else hi = mid; }
    return lo;
}

static void gen_text_chunk(uint8_t *out, size_t n, uint64_t seed, uint64_t chunk_idx) {
    static const char *markup[] = {"<page>","</page>","<title>","</title>","[[","]]","&quot;","==","<id>","</id>","'''","{{","}}","|","* ",
        "<revision>","</revision>","<text xml:space=\"preserve\">","&lt;","&gt;","&amp;","#REDIRECT ","Category:","http://www."};
    const int nmark = sizeof(markup) / sizeof(markup[0]);
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + chunk_idx * 0xD1B54A32D192ED03ull + 1;
    size_t o = 0; int col = 0, wrap = 40 + (int)(rng_next(&s) % 80);
    char tmp[64];
    while (o < n) {
        uint32_t r = (uint32_t)(rng_next(&s) >> 40) & 0xFF;
        const char *src; int L;
        if (r < 14) { src = markup[rng_next(&s) % nmark]; L = (int)strlen(src); }
        else if (r < 20) { L = 1 + (int)(rng_next(&s) % 4); uint64_t v = rng_next(&s); for (int i = 0; i < L; i++) { tmp[i] = (char)('0' + v % 10); v /= 10; } if (L == 4) { tmp[0] = '1' + (tmp[0] & 1); tmp[1] = tmp[0] == '1' ? '9' : '0'; } src = tmp; }
        else { int k = pick_word(&s); src = g_vocab.w[k]; L = g_vocab.len[k]; }
        for (int i = 0; i < L && o < n; i++) out[o++] = (uint8_t)src[i];
        col += L;
        if (o < n) {
            uint32_t q = (uint32_t)(rng_next(&s) >> 33) & 63;
            if (col >= wrap) { out[o++] = '\n'; col = 0; wrap = 40 + (int)(rng_next(&s) % 80); if (q < 8 && o < n) out[o++] = '\n'; }
            else if (q == 0) { out[o++] = ','; if (o < n) out[o++] = ' '; }
            else if (q == 1) { out[o++] = '.'; if (o < n) out[o++] = ' '; }
            else out[o++] = ' ';
        }
    }
}

static void gen_mixed_chunk(uint8_t *out, size_t n, uint64_t seed, uint64_t chunk_idx) {
    uint64_t s = seed * 0xA24BAED4963EE407ull + chunk_idx * 0x9FB21C651E98DF25ull + 7;
    int kind = (int)(rng_next(&s) % 5);
    switch (kind) {
    case 0: gen_text_chunk(out, n, seed ^ 0x55, chunk_idx); break;
    case 1: for (size_t i = 0; i < n; i += 8) { uint64_t v = rng_next(&s); memcpy(out + i, &v, n - i < 8 ? n - i : 8); } break;
    case 2: { int32_t v = (int32_t)rng_next(&s); for (size_t i = 0; i + 4 <= n; i += 4) { v += 3 + (int32_t)(rng_next(&s) & 3); memcpy(out + i, &v, 4); } for (size_t i = n & ~(size_t)3; i < n; i++) out[i] = 0; break; }
    case 3: memset(out, 0, n); for (size_t i = 0; i < n; i += 512 + (rng_next(&s) & 1023)) out[i] = (uint8_t)rng_next(&s); break;
    default: { uint8_t rec[4096]; for (int i = 0; i < 4096; i += 8) { uint64_t v = rng_next(&s); memcpy(rec + i, &v, 8); }
               for (size_t i = 0; i < n; i += 4096) { size_t l = n - i < 4096 ? n - i : 4096; memcpy(out + i, rec, l); if (l > 8) out[i + 4] = (uint8_t)(i >> 12); } break; }
    }
}

typedef struct { uint8_t *out; size_t n; uint64_t seed; int kind; size_t next; size_t nchunks; uint64_t chunk0; pthread_mutex_t mu; } job_t;
static void *worker(void *a) {
    job_t *j = (job_t *)a;
    for (;;) {
        pthread_mutex_lock(&j->mu); size_t c = j->next++; pthread_mutex_unlock(&j->mu);
        if (c >= j->nchunks) break;
        size_t off = c * (size_t)CHUNK, len = j->n - off < CHUNK ? j->n - off : CHUNK;
        if (j->kind == 0) gen_text_chunk(j->out + off, len, j->seed, j->chunk0 + c);
        else gen_mixed_chunk(j->out + off, len, j->seed, j->chunk0 + c);
    }
    return NULL;
}
static void run(uint8_t *out, size_t n, uint64_t seed, uint64_t chunk0, int kind, int nthreads) {
    pthread_once(&g_once, vocab_build);
    job_t j; memset(&j, 0, sizeof j);
    j.out = out; j.n = n; j.seed = seed; j.kind = kind; j.nchunks = (n + CHUNK - 1) / CHUNK; j.chunk0 = chunk0;
    pthread_mutex_init(&j.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    pthread_t t[64];
    for (int i = 0; i < nthreads; i++) pthread_create(&t[i], NULL, worker, &j);
    for (int i = 0; i < nthreads; i++) pthread_join(t[i], NULL);
}
/* chunk0: index of the first 1 MiB chunk (lets rank r generate its own shard of one global stream) */
void w3s_text(uint8_t *out, size_t n, uint64_t seed, uint64_t chunk0, int nthreads) { run(out, n, seed, chunk0, 0, nthreads); }
void w3s_mixed(uint8_t *out, size_t n, uint64_t seed, uint64_t chunk0, int nthreads) { run(out, n, seed, chunk0, 1, nthreads); }
