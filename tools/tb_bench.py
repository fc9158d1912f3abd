"""Time vx_tblock_fused against the three launches it replaces at the 64x64 level (b = 2, f frames, 4096 pixels):
    python tools/tb_bench.py [iters] [f = 16 | 24]"""
import sys

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from v_express_amd import ops

BF = torch.bfloat16


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    b, f, hw, c, heads = 2, (int(sys.argv[2]) if len(sys.argv) > 2 else 16), 4096, 320, 8
    g = torch.Generator().manual_seed(0)
    r = lambda *s, scale=1.0, dt=BF: (torch.randn(*s, generator=g) * scale).to("cuda").to(dt)
    x = r(b * f * hw, c) + 0.3
    wqkv, wo = r(3 * c, c, scale=c ** -0.5), r(c, c, scale=c ** -0.5)
    bq, bo = r(3 * c, dt=torch.float32) * 0.2, r(c, dt=torch.float32) * 0.2
    pe = r(24, 3 * c, dt=torch.float32) * 0.5
    colsum = wqkv.float().sum(dim=1).contiguous()
    rb = pe[:f].repeat(b, 1).contiguous()
    st = torch.empty((b * f * hw, 2), device="cuda")

    def three(h):
        with ops.frame_rows(hw, items=b):
            qkv = ops.gemm(h, wqkv, bq, rowbias=rb, rows_per_group=hw, ln=(st, colsum))
            a = ops.temporal_attention(qkv, b=b, f=f, hw=hw, heads=heads, head_dim=c // heads)
            ops.gemm(a, wo, bo, residual=h, out=h, stats_out=st)

    def fused(h):
        ops.tblock_fused(h, wqkv, bq, colsum, pe, wo, bo, b=b, f=f, hw=hw, heads=heads)

    ops.row_stats(x, out=st)
    for name, fn in (("three launches (qkv + attention + out-proj, stats in / out)", three), ("vx_tblock_fused", fused),
                     ("three launches", three), ("vx_tblock_fused", fused)):
        hs = [x.clone() for _ in range(4)]
        for h in hs:
            fn(h)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(hs[i % 4])
        e1.record()
        torch.cuda.synchronize()
        print(f"f={f} {name:70s} {1e3 * e0.elapsed_time(e1) / iters:8.1f} us")


if __name__ == "__main__":
    main()
