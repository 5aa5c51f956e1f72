"""PCIe-inclusive rate of the host-buffer entry point w3_encode_blocks (pageable numpy buffers), next to the device-resident one.
Run on the GPU box: python3 tools/host_api_rate.py [model] [bytes]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np, torch
    import weath3rb0i_amd as w3
    from tools import synth
    import bench
    name = sys.argv[1] if len(sys.argv) > 1 else "order012apm"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
    bs = 65536
    host = synth.text(n, seed=1)
    model, _ = bench.make_model(w3, name)
    ctx = w3.Context(0)
    out, lens = ctx.encode_blocks(model, host, bs)          # warm-up: workspace allocation
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out, lens = ctx.encode_blocks(model, host, bs); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print("w3_encode_blocks (host buffers, pageable): %.1f ms per call = %.0f MiB/s, ratio %.4f" % (t * 1e3, n / t / 2**20, len(out) / n))
    d_in = torch.from_numpy(host).cuda()
    nb = (n + bs - 1) // bs
    d_out = torch.empty(n + n // 4 + 64 * nb + 1024, dtype=torch.uint8, device="cuda")
    d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda"); d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.encode_blocks_device(model, d_in, bs, d_out, d_lens, d_total); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): ctx.encode_blocks_device(model, d_in, bs, d_out, d_lens, d_total)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
    print("w3_encode_blocks_device (resident): %.1f ms per call = %.0f MiB/s" % (t * 1e3, n / t / 2**20))
    pin = torch.from_numpy(host).pin_memory()
    torch.cuda.synchronize(); t0 = time.perf_counter(); d2 = pin.cuda(non_blocking=True); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("H2D of the input from pinned memory: %.1f ms = %.1f GB/s" % (t * 1e3, n / t / 1e9))


if __name__ == "__main__":
    main()
