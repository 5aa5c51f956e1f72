"""Diagnostic: where does k_partition8 spend its time?  (in-kernel s_memtime stamps per phase, W3_OPT_DEBUG_STAMPS; 100 MHz ticks)"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np, torch
    import weath3rb0i_amd as w3
    from weath3rb0i_amd import _lib as L
    from tools import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
    ctx = w3.Context(0)
    host = synth.text(n, seed=1)
    d_in = torch.from_numpy(host).cuda()
    nb = (n + 65535) // 65536
    d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    for name, m in (("order1: k_partition8<1>", w3.Order1()),):
        ctx.encode_blocks_device(m, d_in, 65536, d_out, d_lens, d_total)
        ctx.lib.w3_ctx_set_option(ctx.h, L.W3_OPT_DEBUG_STAMPS, 1)
        ctx.set_timing(True)
        ctx.encode_blocks_device(m, d_in, 65536, d_out, d_lens, d_total)
        st = (C.c_uint64 * 8)()
        ctx.lib.w3_debug_get_stamps(ctx.h, C.byref(st))
        ctx.lib.w3_ctx_set_option(ctx.h, L.W3_OPT_DEBUG_STAMPS, 0)
        blocks = max(1, st[6])
        names = ["histogram", "tile load + count + scan", "scatter into tile", "(k_rank_sorted, all of it)", "splits", "copy out"]
        print(name, "predict_ms %.2f" % ctx.timing()["predict_ms"], "blocks", st[6])
        for k in range(6):
            print("   %-20s %8.1f us per block" % (names[k], st[k] / blocks / 100.0))
        print("   (rank stamp slot 3 also counts k_rank_sorted's jobs; slot 7 =", st[7], ")")


if __name__ == "__main__":
    main()
