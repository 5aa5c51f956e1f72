// w3_window.h — the input-window loads of the predict kernels, in plain C++ so that the host can run them too: tests/test_window_loads.py
// compiles this file with g++ and drives it over every small block size with the input between two inaccessible pages (a read outside
// the buffer is a SIGSEGV there; on the GPU it is a fault only when the neighbouring page happens to be unmapped — how the reads before
// the buffer of blocks 1, 2 .. at block sizes under 7 bytes stayed unnoticed until round 4).
#pragma once
#include <stdint.h>
#ifndef W3_HD
#define W3_HD __device__ __forceinline__
#endif

// bytes c0..c3 at position i of a block (zeros before the block start: a fresh model's history is 0).
// ONE unconditional unaligned dword load, so that batches of these loads stay in flight together
// (hipcc waits vmcnt(0) right after any load it has to branch around).  Positions 0..2 of a block read into the previous block
// (valid memory) and mask — unless the block starts within three bytes of the input buffer's start (the first block; with blocks of one
// or two bytes also the next ones): `head` = that distance then (W3_NO_HEAD otherwise, the common case: a wave-uniform test), and the
// load starts at the buffer's first byte instead (until round 4 only the first block did that: blocks 1 and 2 of a 1- or 2-byte block
// size read up to two bytes BEFORE the buffer — a fault whenever the page before an allocation was not mapped).
// Needs n >= 4 (the host sends smaller inputs to the generic kernel).
#define W3_NO_HEAD 0xFFFFFFFFu
W3_HD uint32_t window_head(uint64_t off, uint32_t reach) { return off < reach ? (uint32_t)off : W3_NO_HEAD; }
W3_HD uint32_t load_window(const uint8_t *blk, uint32_t i, uint32_t head) {
    // (no branch around the load — see above: the address moves forward by `adj` bytes when the window would start before the buffer, and the
    //  word is shifted back by as many)
    const uint32_t adj = head != W3_NO_HEAD ? 3u - (head + i < 3u ? head + i : 3u) : 0u;
    uint32_t raw;
    __builtin_memcpy(&raw, blk + (int64_t)i - 3 + adj, 4);
    const uint32_t w = __builtin_bswap32(raw) >> (8u * adj);   // memory order c3 c2 c1 c0 -> c0 | c1<<8 | c2<<16 | c3<<24
    const uint32_t sh = 8u * (3u - (i < 3u ? i : 3u));              // bytes of the window before the block start: zeros
    return w & (0xFFFFFFFFu >> sh);
}

// bytes [ic - 7, ic] of a block as one big-endian word (zeros before the block start: a fresh model's history is 0).  `head`: the block
// starts that many bytes (< 7) after the input buffer's start — the window is then taken from the buffer's first eight bytes instead of
// from memory before it (load_window, w3_predict.h) — or W3_NO_HEAD.  Needs n >= 8.
W3_HD uint64_t wave_window(const uint8_t *blk, uint32_t ic, uint32_t head) {
    uint64_t raw, W;
    if (head != W3_NO_HEAD && head + ic < 7u) { __builtin_memcpy(&raw, blk - head, 8); W = __builtin_bswap64(raw) >> (8u * (7u - (head + ic))); }
    else { __builtin_memcpy(&raw, blk + (int64_t)ic - 7, 8); W = __builtin_bswap64(raw); }
    if (ic < 7u) W &= (1ull << (8u * (ic + 1u))) - 1ull;
    return W;
}

