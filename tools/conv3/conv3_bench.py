"""vx_conv3x3_gn (GroupNorm + SiLU applied in the convolution's A path) against the two launches it replaces - the
GroupNorm apply pass into the zero-bordered image + the persistent ring kernel's pad-0 convolution - on the resnet
convolution shapes of a CFG UNet3D forward at 512x512 (32 frames), statistics given by the producer in both cases:

    python tools/conv3/conv3_bench.py [reps]      (after tools/conv3/build_conv3_variants.sh)

Prints us per call of: gn_apply, ring conv, their sum | scale_shift table, conv3_gn, their sum | ratio, and the max
difference between the two results in bf16 ulps of the largest output."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools/conv3")
from v_express_amd import ops  # noqa: E402
import conv3_ops as c3  # noqa: E402

SHAPES = [  # name, frames, H, W, c1, c2, n, residual
    ("L0 conv 320>320 (conv1)", 32, 64, 64, 320, 0, 320, False),
    ("L0 conv 320>320 (conv2 +res)", 32, 64, 64, 320, 0, 320, True),
    ("L0 conv 640cat>320", 32, 64, 64, 320, 320, 320, False),
    ("L0 conv 960cat>320", 32, 64, 64, 640, 320, 320, False),
    ("L1 conv 640>640 (conv2 +res)", 32, 32, 32, 640, 0, 640, True),
    ("L1 conv 320>640 (conv1)", 32, 32, 32, 320, 0, 640, False),
    ("L1 conv 960cat>640", 32, 32, 32, 640, 320, 640, False),
]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    g = torch.Generator().manual_seed(0)
    groups = 32

    def r(*shape, scale=1.0, dtype=torch.bfloat16):
        return (torch.randn(*shape, generator=g) * scale).to("cuda").to(dtype)
    print(f"{'shape':34s} {'gn_apply':>9s} {'ring conv':>10s} {'sum':>8s} | {'table':>7s} {'conv3_gn':>9s} {'sum':>8s} | ratio   TF/s   diff")
    for name, frames, H, W, c1, c2, n, res in SHAPES:
        hw, cin = H * W, c1 + c2
        x1, x2 = r(frames, hw, c1), (r(frames, hw, c2) if c2 else None)
        w = r(n, 9 * cin, scale=(9 * cin) ** -0.5)
        bias, gamma, beta = r(n, dtype=torch.float32), 1 + 0.1 * r(cin, dtype=torch.float32), 0.1 * r(cin, dtype=torch.float32)
        resid = r(frames * hw, n) if res else None
        ws, slices = ops.groupnorm_stats(x1, frames=frames, hw=hw, groups=groups, x2=x2)
        g3 = ops.ConvGeom(frames, H + 2, W + 2, 3, 3, 1, 0)
        buf = {}

        def gn_apply():
            # the apply pass alone over the given statistics (what groupnorm() runs when the producer left them)
            out = ops.padded_buffer(x1.device, frames, H, W, cin)
            ops.L.check(ops._lib.vx_groupnorm_apply(ops._ptr(x1), c1, ops._ptr(x2), c2, frames, hw, groups, 1e-5, ops._ptr(gamma),
                                                    ops._ptr(beta), 1, ops._ptr(out), ops._ptr(ws), slices, ops._gn_slices(hw), W, 1,
                                                    ops._stream()), "vx_groupnorm_apply")
            buf["n"] = out

        def ring():
            with ops.frame_rows(hw, items=2):
                buf["two"] = ops.gemm(buf["n"].view(frames * (H + 2) * (W + 2), -1), w, bias, geom=g3, residual=resid, gn=(groups, hw))

        def table():
            buf["ab"] = c3.groupnorm_scale_shift(ws, slices, gamma, beta, frames=frames, hw=hw, groups=groups, eps=1e-5)

        def conv3():
            p = c3._conv3_params(x1, x2, frames, H, W, n)
            wp = c3.conv3_weight(w)
            out = buf.setdefault("out", torch.empty((frames * hw, n), device="cuda", dtype=torch.bfloat16))
            gws = buf.setdefault("gws", torch.empty((frames, hw // 128, groups, 2), device="cuda"))
            p.w_perm, p.ab, p.silu, p.bias = wp.data_ptr(), buf["ab"].data_ptr(), 1, bias.data_ptr()
            p.out, p.ldc = out.data_ptr(), n
            if resid is not None:
                p.residual, p.ldr = resid.data_ptr(), n
            p.gn_ws, p.gn_groups, p.gn_hw = gws.data_ptr(), groups, hw
            ops.L.check(c3._c3().vx_conv3x3_gn(ops.C.byref(p), ops._stream()), "vx_conv3x3_gn")
        t_a, t_r = timed(gn_apply, reps), timed(ring, reps)
        t_t, t_c = timed(table, reps), timed(conv3, reps)
        d = (buf["out"].float() - buf["two"].float()).abs().max().item()
        ulp = 2.0 ** -8 * buf["two"].float().abs().max().item()
        tf = 2.0 * frames * hw * n * 9 * cin / (t_c * 1e-6) / 1e12
        print(f"{name:34s} {t_a:9.1f} {t_r:10.1f} {t_a + t_r:8.1f} | {t_t:7.1f} {t_c:9.1f} {t_t + t_c:8.1f} | {(t_t + t_c) / (t_a + t_r):5.3f} {tf:6.0f} {d / ulp:6.2f} ulp")


if __name__ == "__main__":
    main()
