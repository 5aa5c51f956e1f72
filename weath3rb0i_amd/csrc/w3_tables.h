// w3_tables.h — host-side generators of the read-only tables the CM kernels stage in LDS.
//   * NaiveStateTable (state_table/naive.rs:9-113): 3963 12-bit states = 3 entry nodes + 4 copies of a
//     990-node triangular count lattice (44 levels).  Written in closed form here:
//     lattice index of (level L, node k) = L(L-1)/2 + k.
//   * stretch / squash: BUILD-DEFINED integer-only logistic pair for the APM stages (the reference has no
//     APM, README.md:10 lists it as a goal; DESIGN.md §2.4 gives the definition).
#pragma once
#include <stdint.h>

namespace w3 {

constexpr int kStLevels = 44;                                  // naive.rs:9  MAX_LEVEL
constexpr int kStLattice = kStLevels * (kStLevels + 1) / 2;    // naive.rs:10 SUBTABLE_SIZE = 990
constexpr int kStSize = 3 + 4 * kStLattice;                    // naive.rs:11 SIZE = 3963

struct StEntry { uint16_t prob, next0, next1, conf; };         // conf: observation-count proxy (replacement policy)

static inline int st_lattice(int level, int k) { return level * (level - 1) / 2 + k; }

static inline void build_state_table(StEntry *t /*[kStSize]*/) {
    const int base[4] = {3, 3 + kStLattice, 3 + 2 * kStLattice, 3 + 3 * kStLattice};   // a, b, c, d  naive.rs:19-22
    t[0] = {32768, 1, 2, 0};                                                            // naive.rs:25
    t[1] = {32768, (uint16_t)base[0], (uint16_t)base[1], 1};                            // naive.rs:26
    t[2] = {32768, (uint16_t)base[2], (uint16_t)base[3], 1};                            // naive.rs:27
    for (int L = 1; L <= kStLevels; L++) {
        for (int k = 0; k < L; k++) {
            const int i = st_lattice(L, k);
            const uint16_t prob = (uint16_t)((65536ull * (uint64_t)(k + 1)) / (uint64_t)(L + 1));   // naive.rs:107-113 (floored)
            int n0, n1;
            if (L < kStLevels) { n0 = st_lattice(L + 1, k); n1 = n0 + 1; }                            // naive.rs:100-103
            else {                                                                                   // naive.rs:81-98: level 44 folds to level 22
                const int tgt = st_lattice(kStLevels / 2, (k + 2) / 2 - 1);
                n0 = k == 0 ? i : tgt;
                n1 = k == L - 1 ? i : tgt;
            }
            for (int c = 0; c < 4; c++) {                                                            // naive.rs:42-45
                const int lo = (c & 1) ? 2 : 0;   // copies a,c continue in (a,b); copies b,d in (c,d)
                t[base[c] + i] = {prob, (uint16_t)(base[lo] + n0), (uint16_t)(base[lo + 1] + n1), (uint16_t)(L + 1)};
            }
        }
    }
}

// squash[d + 2047] = round(2^16 / (1 + e^(-d/256))) clamped to [1, 65535], d in [-2047, 2047];
// e^(-d/256) by repeated Q32 multiplication with K = round(2^32 e^(-1/256)).
// stretch[q] = smallest d with squash(d) >= 16 q + 8 (q = p >> 4), else 2047.
static inline void build_stretch_squash(int16_t *stretch /*[4096]*/, uint16_t *squash /*[4095]*/) {
    const uint64_t K = 0xFF007FD5ull;
    uint64_t e = 1ull << 32;
    for (int d = 0; d <= 2047; d++) {
        const uint64_t den = (1ull << 32) + e;
        uint64_t q = ((1ull << 48) + den / 2) / den;
        if (q > 65535) q = 65535;
        squash[2047 + d] = (uint16_t)q;
        const uint64_t lo = 65536 - q;
        squash[2047 - d] = (uint16_t)(lo < 1 ? 1 : lo);
        e = (e * K) >> 32;
    }
    squash[2047] = 32768;
    int d = -2047;
    for (int q = 0; q < 4096; q++) {
        const uint32_t want = (uint32_t)q * 16u + 8u;
        while (d < 2047 && squash[d + 2047] < want) d++;
        stretch[q] = (int16_t)d;
    }
}

}  // namespace w3
