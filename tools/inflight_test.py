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
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
    bs = 65536; nb = (n + bs - 1) // bs
    name = sys.argv[1] if len(sys.argv) > 1 else "order012apm"
    K = 6 if n >= 500_000_000 else 24
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
    nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    X = [mk() for _ in range(nctx)]
    def run(x, k):
        model, ctx, st, d_out, d_lens, d_total = x
        for _ in range(k):
            ctx.encode_blocks_device(model, d_in, bs, d_out, d_lens, d_total, stream=st.cuda_stream)
    for x in X: run(x, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(X[0], K); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("sequential: %.1f ms/step" % ((t1 - t0) / K * 1e3))
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(x, K // nctx)) for x in X]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    steps = K // nctx * nctx
    print("%d in flight: %.1f ms/step  -> %.0f MiB/s" % (nctx, (t1 - t0) / steps * 1e3, n * steps / (t1 - t0) / 2**20))
    same = all(bool(torch.equal(X[0][3][:int(X[0][5].item())], x[3][:int(x[5].item())])) for x in X[1:])
    print("outputs equal:", same)

if __name__ == "__main__":
    main()
