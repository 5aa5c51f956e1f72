"""Two contexts encoding two batches from two host threads: run as a script on the GPU box."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np, torch
    _run(np, torch)


def _run(np, torch):
    import weath3rb0i_amd as w3
    from tools import synth
    import bench
    n = 1_000_000_000; bs = 65536; nb = (n + bs - 1) // bs
    name = sys.argv[1] if len(sys.argv) > 1 else "order012apm"
    K = 6
    host = synth.text(n, seed=1)
    d_in = torch.from_numpy(host).cuda()
    def mk():
        model, _ = bench.make_model(w3, name)
        ctx = w3.Context(0)
        st = torch.cuda.Stream()
        d_out = torch.empty(n + n // 4 + 64 * nb + 1024, dtype=torch.uint8, device="cuda")
        d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda")
        d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
        return model, ctx, st, d_out, d_lens, d_total
    A = mk(); B = mk()
    def run(X, k):
        model, ctx, st, d_out, d_lens, d_total = X
        for _ in range(k):
            ctx.encode_blocks_device(model, d_in, bs, d_out, d_lens, d_total, stream=st.cuda_stream)
    run(A, 1); run(B, 1); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(A, K); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("sequential: %.1f ms/step" % ((t1 - t0) / K * 1e3))
    t0 = time.perf_counter()
    ta = threading.Thread(target=run, args=(A, K // 2)); tb = threading.Thread(target=run, args=(B, K // 2))
    ta.start(); tb.start(); ta.join(); tb.join(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("two in flight: %.1f ms/step  -> %.0f MiB/s" % ((t1 - t0) / K * 1e3, n * K / (t1 - t0) / 2**20))
    same = bool(torch.equal(A[3][:int(A[5].item())], B[3][:int(B[5].item())]))
    print("outputs equal:", same)


if __name__ == "__main__":
    main()
