"""Debug aid: vx_tblock_fused vs the three launches on one tile, error broken down by pixel / frame / column block."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from v_express_amd import ops
BF = torch.bfloat16
b, hw = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 8
c, heads, f = 320, 8, 16
d = c // heads
g = torch.Generator().manual_seed(0)
r = lambda *s, scale=1.0, dt=BF: (torch.randn(*s, generator=g) * scale).to("cuda").to(dt)
x = r(b * f * hw, c) * 1.5 + 0.3
wqkv, wo = r(3 * c, c, scale=c ** -0.5), r(c, c, scale=c ** -0.5)
bq, bo = r(3 * c, dt=torch.float32) * 0.2, r(c, dt=torch.float32) * 0.2
pe = r(24, 3 * c, dt=torch.float32) * 0.5
colsum = wqkv.float().sum(dim=1).contiguous()
stats = ops.row_stats(x)
with ops.frame_rows(hw, items=b):
    qkv = ops.gemm(x, wqkv, bq, rowbias=pe[:f].repeat(b, 1).contiguous(), rows_per_group=hw, ln=(stats, colsum))
    a = ops.temporal_attention(qkv, b=b, f=f, hw=hw, heads=heads, head_dim=d)
    want = ops.gemm(a, wo, bo, residual=x)
for given in (True, False):
    got = ops.tblock_fused(x.clone(), wqkv, bq, colsum, pe, wo, bo, b=b, f=f, hw=hw, heads=heads, stats=stats if given else None)
    torch.cuda.synchronize()
    e = (got.float() - want.float()).view(b, f, hw, c)
    print(f"stats given={given}: rel-L2 {(e.norm() / want.float().norm()).item():.4g}")
    print(" by pixel (first 16):", [round(v, 3) for v in e.pow(2).mean(dim=(0, 1, 3)).sqrt()[:16].tolist()])
    print(" by frame:", [round(v, 3) for v in e.pow(2).mean(dim=(0, 2, 3)).sqrt().tolist()])
    print(" by col block:", [round(v, 3) for v in e.view(b, f, hw, 20, 16).pow(2).mean(dim=(0, 1, 2, 4)).sqrt().tolist()])
    print(" by item:", [round(v, 3) for v in e.pow(2).mean(dim=(1, 2, 3)).sqrt().tolist()])
# out-projection input check: run the fused kernel with wo = identity-ish to expose O per head
eye = torch.eye(c, device="cuda", dtype=BF)
zero_b = torch.zeros(c, device="cuda")
got = ops.tblock_fused(torch.zeros_like(x) + x, wqkv, bq, colsum, pe, eye, zero_b, b=b, f=f, hw=hw, heads=heads, stats=stats)
o_f = (got.float() - x.float()).view(b, f, hw, heads, d)
o_w = a.float().view(b, f, hw, heads, d)
e = o_f - o_w
print("O (wo = I): rel-L2", (e.norm() / o_w.norm()).item())
print(" by pixel:", [round(v, 3) for v in e.pow(2).mean(dim=(0, 1, 3, 4)).sqrt()[:16].tolist()])
print(" by head:", [round(v, 3) for v in e.pow(2).mean(dim=(0, 1, 2, 4)).sqrt().tolist()])
print(" by channel of head:", [round(v, 3) for v in e.pow(2).mean(dim=(0, 1, 2, 3)).sqrt().tolist()])
print(" by frame:", [round(v, 3) for v in e.pow(2).mean(dim=(0, 2, 3, 4)).sqrt().tolist()])


def probe(name, wq, bqq, peq):
    cs = wq.float().sum(dim=1).contiguous()
    with ops.frame_rows(hw, items=b):
        qkv = ops.gemm(x, wq, bqq, rowbias=peq[:f].repeat(b, 1).contiguous(), rows_per_group=hw, ln=(stats, cs))
        a = ops.temporal_attention(qkv, b=b, f=f, hw=hw, heads=heads, head_dim=d)
    got = ops.tblock_fused(x.clone(), wq, bqq, cs, peq, eye, zero_b, b=b, f=f, hw=hw, heads=heads, stats=stats)
    torch.cuda.synchronize()
    e = ((got.float() - x.float()) - a.float()).view(b, f, hw, heads, d)
    print(f"probe {name}: O rel-L2 {(e.norm() / a.float().norm()).item():.4g}; by pixel",
          [round(v, 3) for v in e.pow(2).mean(dim=(0, 1, 3, 4)).sqrt()[:8].tolist()],
          "by channel block", [round(v, 3) for v in e.view(b, f, hw, heads, 5, 8).pow(2).mean(dim=(0, 1, 2, 3, 5)).sqrt().tolist()])


col = torch.arange(3 * c, device="cuda")
part, dd = col // c, (col % c) % d
# (a) q = 0: uniform softmax, O = mean over frames of V
m = (part != 0).to(BF)
probe("q = 0 (uniform softmax)", wqkv * m[:, None], bq * m.float(), pe * m.float())
# (b) channels 32..39 of q zero: the mixed block contributes nothing
m = (~((part == 0) & (dd >= 32))).to(BF)
probe("q[32:40] = 0 (no mixed-block term)", wqkv * m[:, None], bq * m.float(), pe * m.float())
# (c) channels 0..31 of q zero: ONLY the mixed block contributes
m = (~((part == 0) & (dd < 32))).to(BF)
probe("q[0:32] = 0 (only the mixed-block term)", wqkv * m[:, None], bq * m.float(), pe * m.float())
# (d) no bias / pe at all
probe("no bias, no pe", wqkv, bq * 0, pe * 0)
