"""Time the 16x16-level launches with and without the cooperative two-way K split of the persistent kernel
(vx_gemm_params.ring_hint = 2, ops.RING_COOP):   python tools/coop_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v_express_amd import ops  # noqa: E402

BF = torch.bfloat16


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    g = torch.Generator().manual_seed(0)
    r = lambda *s, scale=1.0, dt=BF: (torch.randn(*s, generator=g) * scale).to("cuda").to(dt)
    nb, hh, ww = 32, 16, 16
    m = nb * hh * ww
    print(f"{'launch':46s} {'classic us':>10s} {'TF/s':>7s} {'coop us':>9s} {'TF/s':>7s}  kernels")
    cases = [("linear", 1280, 5120), ("linear", 1280, 2560), ("linear", 1280, 1280), ("conv", 1280, 1280), ("conv", 1280, 2560),
             ("conv", 1280, 640), ("conv", 1280, 1920)]
    for kind, n, c in cases:
        bias = r(n, dt=torch.float32)
        res = r(m, n)
        if kind == "linear":
            a, w = r(m, c), r(n, c, scale=c ** -0.5)
            k = c
            fn = lambda: ops.gemm(a, w, bias, residual=res)
        else:
            x = torch.zeros(nb, hh + 2, ww + 2, c, device="cuda", dtype=BF)
            x[:, 1:-1, 1:-1] = r(nb, hh, ww, c)
            w = r(n, 9 * c, scale=(9 * c) ** -0.5)
            k = 9 * c
            geom = ops.ConvGeom(nb, hh + 2, ww + 2, 3, 3, 1, 0)
            xv = x.view(-1, c)
            fn = lambda: ops.gemm(xv, w, bias, geom=geom, residual=res, gn=(32, hh * ww))
        t, names = [], []
        for coop in (False, True):
            ops.RING_COOP[0] = coop
            with ops.frame_rows(hh * ww, items=2):
                with ops.GemmProfile() as prof:
                    fn()
                names.append(prof.records[0][3].split("<")[0] + ("+coop" if prof.records[0][3].endswith(",coop2>") else ""))
                t.append(timed(fn, iters))
        fl = 2.0 * m * n * k
        print(f"{kind:6s} {m:6d} x {n:5d} x {k:6d} (+res{', GN sums' if kind == 'conv' else ''}){'':6s} {t[0]:10.1f} {fl / t[0] / 1e6:7.1f} "
              f"{t[1]:9.1f} {fl / t[1] / 1e6:7.1f}  {names[0]} | {names[1]}")
    ops.RING_COOP[0] = True


if __name__ == "__main__":
    main()
