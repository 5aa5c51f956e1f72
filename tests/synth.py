"""Seeded synthetic inputs shared by the parity tests (no corpora exist here)."""
import numpy as np

ALPH = b"etaoin shrdlu<>/=\"[]&;\n0123456789ETAOIN"


def lcg_text(n, seed=12345):
    """SURVEY §8(c) cross-check generator: x=(x*1664525+1013904223) mod 2^32; ALPH[(x>>24)%39]."""
    assert len(ALPH) == 39
    a, c = np.uint64(1664525), np.uint64(1013904223)
    out = np.empty(n, dtype=np.uint8)
    x = np.uint64(seed)
    m = np.uint64(0xFFFFFFFF)
    alph = np.frombuffer(ALPH, dtype=np.uint8)
    # vectorised in chunks via the closed form x_{k} = A^k x + C_k is overkill; a plain loop is fine for <=1 MiB
    xi = int(seed)
    buf = bytearray(n)
    for i in range(n):
        xi = (xi * 1664525 + 1013904223) & 0xFFFFFFFF
        buf[i] = ALPH[(xi >> 24) % 39]
    return bytes(buf)


def markov_text(n, seed=1):
    """Cheap enwik-ish text: Zipf-weighted words + XML-ish tokens (numpy, deterministic)."""
    rng = np.random.default_rng(seed)
    syll = ["th", "e", "an", "in", "er", "on", "re", "ed", "st", "ar", "ti", "al", "or", "is", "at", "en", "le",
            "ou", "ng", "co", "de", "ra", "li", "ic", "ion", "ent", "pro", "qu", "mp", "ly"]
    words = []
    for _ in range(2000):
        k = int(rng.integers(1, 5))
        words.append("".join(syll[int(j)] for j in rng.integers(0, len(syll), k)))
    words += ["<page>", "</page>", "<title>", "</title>", "[[", "]]", "&quot;", "==", "<id>", "</id>", "\n", "\n\n",
              "1999", "2004", "'''", "{{", "}}", "|", "*", "#REDIRECT"]
    w = 1.0 / np.arange(1, len(words) + 1) ** 1.05
    w /= w.sum()
    out = bytearray()
    while len(out) < n:
        idx = rng.choice(len(words), size=4096, p=w)
        for j in idx:
            out += words[int(j)].encode()
            out += b" "
    return bytes(out[:n])


def mixed_bytes(n, seed=7):
    """Silesia-ish: text, random, zero runs, ramps, repeated records."""
    rng = np.random.default_rng(seed)
    parts = []
    left = n
    kinds = 0
    while left > 0:
        m = int(min(left, rng.integers(1, 9) * 4096))
        k = kinds % 5
        kinds += 1
        if k == 0:
            parts.append(markov_text(m, seed + kinds))
        elif k == 1:
            parts.append(rng.integers(0, 256, m, dtype=np.uint8).tobytes())
        elif k == 2:
            parts.append(bytes(m))
        elif k == 3:
            r = (np.arange(m // 4 + 1, dtype=np.int32) * 3 + rng.integers(0, 4, m // 4 + 1).astype(np.int32))
            parts.append(r.tobytes()[:m])
        else:
            rec = rng.integers(0, 256, 256, dtype=np.uint8).tobytes()
            parts.append((rec * (m // 256 + 1))[:m])
        left -= m
    return b"".join(parts)[:n]
