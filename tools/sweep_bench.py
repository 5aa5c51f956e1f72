"""Times the one-launch OrderN sweep (weath3rb0i_amd/sweep.py, bin/ordern/main.rs) on 20 MB of the synthetic text: run as a script on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import weath3rb0i_amd as w3
    from weath3rb0i_amd import sweep
    from tools import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    data = synth.text(n, seed=1)
    ctx = w3.Context(0)
    lines = []
    t0 = time.perf_counter()
    best, params, table = sweep.sweep_ordern(ctx, data, 65536, range(8, 31), range(0, 5), repeats=1, out=lines.append)
    dt = time.perf_counter() - t0
    print("\n".join(lines[-8:]))
    print("sweep of %d configurations over %d bytes (%d blocks of 64 KiB): %.2f s" % (len(table), n, (n + 65535) // 65536, dt))
    ctx.close()


if __name__ == "__main__":
    main()
