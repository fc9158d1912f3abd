"""Does a small-M linear pay for COLD weights?  The 16x16 / 8x8-level shapes (M = 8192 / 4096 / 2048, N = K = 1280) timed with
(a) one weight tensor reused by every launch (L2-warm, what a stand-alone benchmark measures), (b) a rotation over 64 weight
tensors = 210 MB (every launch streams its weights from HBM / Infinity Cache, what the model does: 1.7 GB of weights per step),
(c) = (b) with a rotation of the activations as well.  GPU box: python tools/weight_cold_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import v_express_amd as vx  # noqa: F401
    from v_express_amd import ops
    dev = torch.device("cuda", 0)
    reps = 200
    for m, n, k, hw in ((8192, 1280, 1280, 256), (4096, 1280, 1280, 256), (2048, 1280, 1280, 64), (8192, 1280, 5120, 256),
                        (32768, 640, 640, 1024), (131072, 320, 320, 4096)):
        nw = 64
        ws = [(torch.randn(n, k, device=dev) * k ** -0.5).to(ops.BF16) for _ in range(nw)]
        na = max(2, min(32, int(4e8 // (m * k * 2))))
        xs = [torch.randn(m, k, device=dev).to(ops.BF16) for _ in range(na)]
        rs = [torch.randn(m, n, device=dev).to(ops.BF16) for _ in range(na)]
        bias = torch.randn(n, device=dev)
        out = torch.empty(m, n, device=dev, dtype=ops.BF16)
        res = {}
        for name, wsel, asel in (("warm weights", lambda i: 0, lambda i: 0), ("cold weights", lambda i: i % nw, lambda i: 0),
                                 ("cold weights + activations", lambda i: i % nw, lambda i: i % na)):
            with ops.frame_rows(hw, items=2):
                for i in range(10):
                    ops.gemm(xs[asel(i)], ws[wsel(i)], bias, residual=rs[asel(i)], out=out)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    ops.gemm(xs[asel(i)], ws[wsel(i)], bias, residual=rs[asel(i)], out=out)
                e1.record()
            torch.cuda.synchronize()
            res[name] = 1e3 * e0.elapsed_time(e1) / reps
        print(f"{m:7d} x {n:5d} x {k:5d}: " + "   ".join(f"{a} {b:6.1f} us" for a, b in res.items()), flush=True)
        del ws, xs, rs


if __name__ == "__main__":
    main()
