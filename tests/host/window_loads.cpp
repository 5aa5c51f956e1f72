// Host harness for weath3rb0i_amd/csrc/w3_window.h (tests/test_window_loads.py): the predict kernels' input-window loads, compiled for
// the CPU and driven over every small block size with the input buffer flush against an inaccessible page on either side — a read
// outside [buf, buf + n) is a SIGSEGV here, and every value is compared with the definition (zeros before the block start).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <unistd.h>

#define W3_HD static inline
namespace w3 {
#include "../../weath3rb0i_amd/csrc/w3_window.h"
}

static uint32_t want4(const uint8_t *blk, uint32_t i) {          // c0 | c1 << 8 | c2 << 16 | c3 << 24, c_k = byte i - k of the block or 0
    uint32_t w = 0;
    for (uint32_t k = 0; k < 4 && k <= i; k++) w |= (uint32_t)blk[i - k] << (8 * k);
    return w;
}
static uint64_t want8(const uint8_t *blk, uint32_t ic) {         // bytes [ic - 7, ic] big-endian, zeros before the block start
    uint64_t w = 0;
    for (uint32_t k = 0; k < 8 && k <= ic; k++) w |= (uint64_t)blk[ic - k] << (8 * k);
    return w;
}

int main() {
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    uint8_t *m = (uint8_t *)mmap(nullptr, 3 * page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) { perror("mmap"); return 2; }
    if (mprotect(m, page, PROT_NONE) || mprotect(m + 2 * page, page, PROT_NONE)) { perror("mprotect"); return 2; }
    uint8_t *lo = m + page, *hi = m + 2 * page;                  // the accessible page
    unsigned long checks = 0;
    for (int at_end = 0; at_end < 2; at_end++)                   // buffer flush with the page's start (reads before it trap) / with its end (reads past it trap)
        for (size_t n = 8; n <= 48; n++) {
            uint8_t *buf = at_end ? hi - n : lo;
            for (size_t k = 0; k < n; k++) buf[k] = (uint8_t)(0x11 + 37 * k + 3 * n);
            for (size_t bs = 1; bs <= 12; bs++)
                for (size_t off = 0; off < n; off += bs) {
                    const uint8_t *blk = buf + off;
                    const uint32_t len = (uint32_t)std::min(bs, n - off);
                    const uint32_t h3 = w3::window_head(off, 3u), h7 = w3::window_head(off, 7u);
                    for (uint32_t i = 0; i < len; i++) {
                        const uint32_t g4 = w3::load_window(blk, i, h3);
                        const uint64_t g8 = w3::wave_window(blk, i, h7);
                        if (g4 != want4(blk, i) || g8 != want8(blk, i)) {
                            printf("MISMATCH n=%zu bs=%zu off=%zu i=%u: %08x / %08x, %016llx / %016llx\n", n, bs, off, i, g4, want4(blk, i),
                                   (unsigned long long)g8, (unsigned long long)want8(blk, i));
                            return 1;
                        }
                        checks++;
                    }
                }
        }
    printf("window loads ok: %lu positions\n", checks);
    return 0;
}
