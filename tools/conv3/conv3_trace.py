"""Slot timing of vx_conv3x3_gn from the trace build (tools/build_conv3_variants.sh "trace:-DVX_C3_TRACE"):

    VX_LIBRARY=$PWD/tools/c3libs/trace.so python tools/conv3_trace.py

Waves 0 (wave row 0) and 4 (wave row 1) of block 0 stamp the cycle counter (100 MHz "realtime"-independent shader clock
counter of s_memtime) at four points of every phase: end of the L slot's own work, after its barrier, end of the M slot's
MFMA issue, after its barrier.  Printed per position v of the nine-K-tile period and phase: L work, L barrier wait, M work,
M barrier wait (average over the traced K-tiles, in counter ticks), and the K-tile total."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools/conv3")
from v_express_amd import ops  # noqa: E402
import conv3_ops as c3  # noqa: E402


def main():
    g = torch.Generator().manual_seed(0)

    def r(*shape, scale=1.0, dtype=torch.bfloat16):
        return (torch.randn(*shape, generator=g) * scale).to("cuda").to(dtype)
    frames, H, W, c, n, groups = 32, 64, 64, 320, 320, 32
    hw = H * W
    x = r(frames, hw, c)
    w = r(n, 9 * c, scale=(9 * c) ** -0.5)
    bias, gamma, beta = r(n, dtype=torch.float32), 1 + 0.1 * r(c, dtype=torch.float32), 0.1 * r(c, dtype=torch.float32)
    res = r(frames * hw, n)
    buf = torch.zeros(2 * 513, device="cuda", dtype=torch.int64)
    fn = c3._c3().vx_conv3_set_trace
    fn.argtypes = [ctypes.c_void_p]
    for _ in range(3):
        assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
        c3.conv3_gn(x, gamma, beta, w, bias, frames=frames, H=H, W=W, groups=groups, eps=1e-5, residual=res, gn=(groups, hw))
    torch.cuda.synchronize()
    t = buf.cpu().view(2, 513)
    for row in (0, 1):
        nst = int(t[row, 0])
        st = t[row, 1:1 + nst].tolist()
        print(f"wave row {row}: {nst} stamps")
        # 16 stamps per K-tile: phases 0..3 x [L end, after L barrier, M end, after M barrier]
        per = {}
        nkt = nst // 16
        prev = None
        for kt in range(nkt):
            s16 = st[16 * kt:16 * kt + 16]
            v = kt % 9
            for ph in range(4):
                le, lb, me, mb = s16[4 * ph:4 * ph + 4]
                lw = (le - prev) if prev is not None else 0
                per.setdefault((v, ph), []).append((lw, lb - le, me - lb, mb - me))
                prev = mb
        print("  v ph |   L work  L wait |   M work  M wait")
        tot_all = 0.0
        for v in range(9):
            tot = 0.0
            for ph in range(4):
                a = per.get((v, ph), [])
                if not a:
                    continue
                m = [sum(x[i] for x in a) / len(a) for i in range(4)]
                tot += sum(m)
                print(f"  {v}  {ph} | {m[0]:8.0f} {m[1]:7.0f} | {m[2]:8.0f} {m[3]:7.0f}")
            print(f"  K-tile v={v}: {tot:8.0f} ticks")
            tot_all += tot
        print(f"  period: {tot_all:.0f} ticks; first / last stamp span {st[-1] - st[0]} over {nkt} K-tiles")


if __name__ == "__main__":
    main()
