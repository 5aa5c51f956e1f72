// w3_huff.h — host-side table preparation of HuffHistory (history/huff_history.rs:17-55): length-limited Huffman code
// lengths by package-merge (entropy_coding/package_merge.rs:1-84), canonical codes (:87-117), bit reversal.
// Constructor-time work on 256 symbols; the per-bit hash runs on the device (w3_predict.h k_huffkeys, w3_generic.h).
// Equal keys are taken in ascending symbol order (std::stable_sort); the reference's sort_unstable_by leaves that
// order to Rust's sort implementation.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/w3hip.h"

namespace w3huff {

// code lengths for counts sorted ascending (package_merge_sorted, :34-84): `levels[d]` = the merged list of depth d, each
// item either an original symbol or a package of the two items before it in the previous level
static inline std::vector<uint8_t> lengths_sorted(const std::vector<uint64_t> &a, unsigned max_len) {
    const size_t n = a.size();
    std::vector<uint8_t> lens(n, 0);
    if (n < 2) return lens;
    const size_t cap = 2 * n - 1;
    std::vector<uint32_t> packaged(cap, 0);   // bit d set: the item at this rank of level d is a package
    std::vector<uint64_t> prev(a), cur;
    for (unsigned d = 1; d < max_len; d++) {
        cur.clear();
        size_t s = 0, p = 0;
        const size_t np = prev.size() / 2;
        while (p < np || s < n) {
            const bool take_pkg = p < np && (s >= n || prev[2 * p] + prev[2 * p + 1] <= a[s]);
            if (take_pkg) { packaged[cur.size()] |= 1u << d; cur.push_back(prev[2 * p] + prev[2 * p + 1]); p++; }
            else cur.push_back(a[s++]);
        }
        prev.swap(cur);
    }
    size_t relevant = 2 * n - 2;
    for (int d = (int)max_len - 1; d >= 0 && relevant; d--) {
        size_t sym = 0;
        for (size_t i = 0; i < relevant && i < cap; i++)
            if (!(packaged[i] >> d & 1u)) lens[sym++] += 1;
        relevant = (relevant - sym) * 2;
    }
    return lens;
}

// package_merge (:1-29); false = one of the reference's asserts would fire
static inline bool code_lengths(const uint32_t *counts, size_t n, unsigned max_len, uint8_t *out) {
    std::vector<std::pair<uint32_t, uint32_t>> sc;   // (count, symbol)
    for (size_t i = 0; i < n; i++)
        if (counts[i]) sc.emplace_back(counts[i], (uint32_t)i);
    std::stable_sort(sc.begin(), sc.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
    if (sc.empty() || max_len > 32 || (max_len < 32 && sc.size() > ((size_t)1 << max_len))) return false;
    std::vector<uint64_t> a(sc.size());
    for (size_t i = 0; i < sc.size(); i++) a[i] = sc[i].first;
    const std::vector<uint8_t> l = lengths_sorted(a, max_len);
    memset(out, 0, n);
    for (size_t i = 0; i < sc.size(); i++) out[sc[i].second] = l[i];
    return true;
}

// canonical (:87-117) followed by HuffHistory::new's bit reversal (huff_history.rs:21-25)
static inline void reversed_canonical(const uint8_t *lens, size_t n, uint16_t *code, uint8_t *len) {
    std::vector<std::pair<uint8_t, uint32_t>> sl;
    unsigned max_len = 0;
    for (size_t i = 0; i < n; i++) {
        max_len = std::max<unsigned>(max_len, lens[i]);
        if (lens[i]) sl.emplace_back(lens[i], (uint32_t)i);
    }
    std::stable_sort(sl.begin(), sl.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
    std::vector<uint16_t> cnt(max_len + 2, 0), next(max_len + 2, 0);
    for (const auto &e : sl) cnt[e.first]++;
    for (unsigned i = 0; i < max_len; i++) next[i + 1] = (uint16_t)((next[i] + cnt[i]) << 1);
    for (size_t i = 0; i < n; i++) { code[i] = 0; len[i] = 0; }
    for (const auto &e : sl) {
        const uint16_t c = next[e.first]++;
        uint16_t r = 0;
        for (int b = 0; b < 16; b++) r = (uint16_t)((r << 1) | ((c >> b) & 1));
        code[e.second] = (uint16_t)(r >> ((16u - e.first) & 15u));   // reverse_bits().overflowing_shr(16 - len)
        len[e.second] = e.first;
    }
}

static inline bool build(const uint8_t *buf, size_t n, unsigned huff_size, unsigned rem_size, w3_huff_table *out) {
    uint32_t counts[256] = {0}, rem[256] = {0};
    uint8_t lens[256];
    for (size_t i = 0; i < n; i++) counts[buf[i]]++;
    if (!code_lengths(counts, 256, huff_size, lens)) return false;
    reversed_canonical(lens, 256, out->code, out->len);
    for (int byte = 0; byte < 256; byte++)                    // huff_history.rs:27-34
        for (int bl = 0; bl < 8; bl++) rem[(1 << bl) | (byte >> (8 - bl))] += counts[byte];
    if (!code_lengths(rem, 256, rem_size, lens)) return false;
    reversed_canonical(lens, 256, out->rem_code, out->rem_len);
    return true;
}

}  // namespace w3huff
