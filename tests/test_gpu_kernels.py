"""Per-kernel parity on a real MI355X: every libvexpress_hip entry point, called through the C ABI (ctypes), against
a float32 PyTorch restatement of the same op fed the same bf16-rounded inputs.

Tolerance (written per test): outputs are bf16, so the only error vs an fp32 evaluation of the same bf16 inputs is
accumulation order + one final rounding:  max|err| <= 2^-7 * max|ref| (+ tiny abs) and relative L2 <= 6e-3.
"""

import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# VX_TEST_ELEM=f16 runs this whole file against the IEEE-half build of the library (libvexpress_hip_f16.so, the reference's
# default --dtype fp16): inputs are float16, every call goes to that library, and every tolerance is 8x tighter (11 instead of
# 8 mantissa bits: max|err| <= 2^-10 max|ref|, relative L2 <= 7.5e-4) - except the fp8 tests, whose error is the e4m3 operands'.
# The default run (bfloat16) is what the driver executes; tests/test_gpu_f16.py holds the float16 cases that always run.
F16 = os.environ.get("VX_TEST_ELEM", "") == "f16"
BF = torch.float16 if F16 else torch.bfloat16
TOL = 0.125 if F16 else 1.0


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from v_express_amd import lib as L, ops as o
    with L.element_type(BF):
        yield o


def rnd(*shape, scale=1.0, seed=0, dtype=BF):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to("cuda").to(dtype)


def check(got, ref, what, rel=6e-3, mx=2 ** -7):
    if "fp8" not in what:
        rel, mx = rel * TOL, mx * TOL
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{what}: non-finite output ({(~torch.isfinite(got)).sum().item()} elements)"
    err = (got - ref).abs()
    scale = ref.abs().max().item() + 1e-12
    rl2 = (err.pow(2).sum().sqrt() / (ref.pow(2).sum().sqrt() + 1e-12)).item()
    i = err.argmax().item()
    idx = tuple(int(v) for v in torch.unravel_index(torch.tensor(i), got.shape))
    msg = (f"{what}: max|err|={err.max().item():.4g} at {idx} (got {got.flatten()[i].item():.5g}, ref "
           f"{ref.flatten()[i].item():.5g}), max|ref|={scale:.4g}, relL2={rl2:.3g}; "
           f"frac>tol={(err > mx * scale + 1e-5).float().mean().item():.4g}")
    assert err.max().item() <= mx * scale + 1e-5 and rl2 <= rel, msg


# ----------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("m,n,k", [(256, 320, 320), (200, 64, 72), (131, 1280, 640), (64, 8, 2880), (2, 1280, 320),
                                   (300, 192, 64), (1000, 960, 320), (257, 128, 128), (512, 2560, 1280)])
def test_gemm_plain(ops, m, n, k):
    a, w = rnd(m, k), rnd(n, k, scale=k ** -0.5, seed=1)
    bias = rnd(n, seed=2, dtype=torch.float32)
    out = ops.gemm(a, w, bias)
    check(out, a.float() @ w.float().t() + bias, f"gemm {m}x{n}x{k}")


def test_gemm_split_k(ops):
    """8x8-level problems run split-K (slices summed in slice order by a second launch): parity with the fp32
    reference, and a half-batch launch is bit-identical to the matching rows of the full one (the split depends on the
    per-frame geometry, N and K only)."""
    from v_express_amd import lib as L
    nb, H, W, cin, cout = 6, 8, 8, 320, 320            # K = 2880 (>= 2560: the split-K side of the policy)
    x = rnd(nb, H + 2, W + 2, cin)
    wt = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=1)
    bias = rnd(cout, seed=2, dtype=torch.float32)
    res = rnd(nb * H * W, cout, seed=3)
    w2d = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    g = ops.ConvGeom(nb, H + 2, W + 2, 3, 3, 1, 0)
    with ops.GemmProfile() as prof:
        out = ops.gemm(x.view(-1, cin), w2d, bias, geom=g, residual=res, alpha=0.9, act=L.VX_ACT_SILU)
    assert "splitk" in prof.records[0][3], prof.records[0][3]
    ref = F.silu(_conv_ref(x, wt, bias, 1, 0, 0)).reshape(nb * H * W, cout) * 0.9 + res.float()
    check(out, ref, "split-K conv + silu + residual")
    half = ops.gemm(x[:3].reshape(-1, cin), w2d, bias, geom=ops.ConvGeom(3, H + 2, W + 2, 3, 3, 1, 0),
                    residual=res[:3 * H * W], alpha=0.9, act=L.VX_ACT_SILU)
    assert torch.equal(half, out[:3 * H * W])
    # plain linear under the frame_rows hint, f32 output: K >= 2560 splits ...
    a, w = rnd(2 * 64, 2560), rnd(320, 2560, scale=2560 ** -0.5, seed=4)
    with ops.frame_rows(64), ops.GemmProfile() as prof:
        o2 = ops.gemm(a, w, bias, out_f32=True)
    assert "splitk" in prof.records[0][3]
    check(o2, a.float() @ w.float().t() + bias, "split-K linear f32", rel=1e-4, mx=1e-4)
    # ... K < 2560 runs unsplit on the 64 x 160 two-wave tile (bf16 output; f32 output keeps the 128-row tiles)
    a, w = rnd(32 * 64, 1280), rnd(1280, 1280, scale=1280 ** -0.5, seed=5)
    b2, r2 = rnd(1280, seed=6, dtype=torch.float32), rnd(32 * 64, 1280, seed=7)
    with ops.frame_rows(64), ops.GemmProfile() as prof:
        o3 = ops.gemm(a, w, b2, residual=r2)
    assert "splitk" not in prof.records[0][3] and "64x160" in prof.records[0][3], prof.records[0][3]
    check(o3, r2.float() + a.float() @ w.float().t() + b2, "8x8-level linear on the 64x160 tile")
    with ops.frame_rows(64):
        half3 = ops.gemm(a[:16 * 64], w, b2, residual=r2[:16 * 64])
    assert torch.equal(half3, o3[:16 * 64])


def _ring_used(prof):
    return all("gemm_ring" in r[3] for r in prof.records) and len(prof.records) > 0


@pytest.mark.parametrize("m,n,k", [(256 * 96, 640, 320), (256 * 192, 320, 64), (256 * 200, 320, 128),
                                   (256 * 300, 320, 192), (256 * 64, 960, 1280)])
def test_gemm_ring_linear(ops, m, n, k):
    """Persistent ring-staged kernel (vx_gemm_ring.hip): plain linears incl. 1- and 2-K-tile problems and blocks that
    walk more than one output tile; bias + alpha + in-place residual through the 16-byte permlane epilogue."""
    a, w = rnd(m, k), rnd(n, k, scale=k ** -0.5, seed=1)
    bias = rnd(n, seed=2, dtype=torch.float32)
    res = rnd(m, n, seed=3)
    h = res.clone()
    with ops.GemmProfile() as prof:
        ops.gemm(a, w, bias, residual=h, alpha=0.75, out=h)
    assert _ring_used(prof), prof.records[0][3]
    check(h, res.float() + 0.75 * (a.float() @ w.float().t() + bias), f"ring gemm {m}x{n}x{k}")
    # a launch over the first half of the rows is bit-identical to the matching rows (K order is shape-independent)
    if (m // 2) % 256 == 0 and (m // 2 // 256) * (n // 320) >= 192:
        h2 = res[:m // 2].clone()
        ops.gemm(a[:m // 2], w, bias, residual=h2, alpha=0.75, out=h2)
        assert torch.equal(h2, h[:m // 2])


@pytest.mark.parametrize("m,c", [(256 * 24, 320), (256 * 13, 640)])
def test_gemm_ring_geglu(ops, m, c):
    """GEGLU epilogue of the ring kernel (8-row value/gate interleave, permlane32 value/gate pairing, 16-byte stores)."""
    from v_express_amd import weights as Wt
    a = rnd(m, c)
    w = rnd(8 * c, c, scale=c ** -0.5, seed=1)
    b = rnd(8 * c, seed=2, dtype=torch.float32)
    with ops.GemmProfile() as prof:
        out = ops.geglu(a, Wt.geglu_interleave(w), Wt.geglu_interleave(b))
    assert _ring_used(prof), prof.records[0][3]
    hg = a.float() @ w.float().t() + b
    hval, gate = hg.chunk(2, dim=-1)
    check(out, hval * F.gelu(gate), f"ring geglu c={c}")


def test_gemm_ring_store_no_residual(ops):
    """STORE epilogue without a residual (separate kernel instantiation) + per-tile time-embedding row."""
    m, n, k, grp = 256 * 96, 640, 192, 256 * 48
    a, w = rnd(m, k), rnd(n, k, scale=k ** -0.5, seed=1)
    bias = rnd(n, seed=2, dtype=torch.float32)
    rowbias = rnd(2, n, seed=3, dtype=torch.float32)
    with ops.GemmProfile() as prof:
        out = ops.gemm(a, w, bias, rowbias=rowbias, rows_per_group=grp)
    assert _ring_used(prof), prof.records[0][3]
    check(out, a.float() @ w.float().t() + bias + rowbias.repeat_interleave(grp, 0), "ring gemm bias+rowbias")


@pytest.mark.parametrize("nb,hh,ww,c1,c2,cout", [(48, 32, 32, 64, 0, 320), (24, 32, 32, 64, 128, 640),
                                                 (768, 8, 8, 64, 0, 320), (192, 16, 16, 64, 64, 320)])
def test_gemm_ring_conv(ops, nb, hh, ww, c1, c2, cout):
    """3x3 convolution over a zero-bordered image (the resnet path) through the ring kernel: tap-innermost K order,
    dual-source channel concat, time-embedding rows + SiLU; tiles spanning image rows (32x32), one frame (16x16) and
    four frames (8x8)."""
    from v_express_amd import lib as L
    cin = c1 + c2
    x = torch.zeros(nb, hh + 2, ww + 2, cin, device="cuda", dtype=BF)
    x[:, 1:-1, 1:-1] = rnd(nb, hh, ww, cin)
    wt = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=1)
    bias = rnd(cout, seed=2, dtype=torch.float32)
    w2d = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    g = ops.ConvGeom(nb, hh + 2, ww + 2, 3, 3, 1, 0)
    rows = hh * ww * (nb // 2)
    rowbias = rnd(2, cout, seed=5, dtype=torch.float32)
    if c2:
        x1, x2 = x[..., :c1].contiguous(), x[..., c1:].contiguous()
        a, a2 = x1.view(-1, c1), x2.view(-1, c2)
    else:
        a, a2 = x.view(-1, cin), None
    with ops.GemmProfile() as prof:
        out = ops.gemm(a, w2d, bias, geom=g, a2=a2, rowbias=rowbias, rows_per_group=rows, act=L.VX_ACT_SILU)
    assert _ring_used(prof), prof.records[0][3]
    ref = _conv_ref(x, wt, bias, 1, 0, 0).reshape(nb * hh * ww, cout) + rowbias.repeat_interleave(rows, 0)
    check(out, F.silu(ref), f"ring conv {nb}x{hh}x{ww} {c1}+{c2}->{cout}")


def test_gemm_epilogue_options(ops):
    from v_express_amd import lib as L
    m, n, k, grp = 384, 320, 256, 96
    a, w = rnd(m, k), rnd(n, k, scale=k ** -0.5, seed=1)
    bias = rnd(n, seed=2, dtype=torch.float32)
    rowbias_full = rnd(m // grp, 3 * n, seed=3, dtype=torch.float32)
    rowbias = rowbias_full[:, n:2 * n]                       # strided view, like the time-embedding slices
    res = rnd(m, n, seed=4)
    base = a.float() @ w.float().t() + bias + rowbias.repeat_interleave(grp, 0)
    out = ops.gemm(a, w, bias, rowbias=rowbias, rows_per_group=grp, residual=res, alpha=0.95)
    check(out, res.float() + 0.95 * base, "gemm bias+rowbias+alpha+residual")
    out = ops.gemm(a, w, bias, act=L.VX_ACT_SILU, out_f32=True)
    assert out.dtype == torch.float32
    check(out, F.silu(a.float() @ w.float().t() + bias), "gemm silu f32-out", rel=1e-4, mx=1e-4)
    # in-place residual (out aliases residual) and strided A / out views
    big = rnd(m, 3 * k, seed=5)
    h = res.clone()
    ops.gemm(big[:, k:2 * k], w, bias, residual=h, out=h)
    check(h, res.float() + big[:, k:2 * k].float() @ w.float().t() + bias, "gemm in-place residual, strided A")


def _conv_ref(x_nhwc, w_oihw, bias, stride, pad, upsample):
    x = x_nhwc.float().permute(0, 3, 1, 2)
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    y = F.conv2d(x, w_oihw.float(), bias, stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("nb,h,w,cin,cout,kk,stride,pad,ups", [
    (3, 16, 16, 64, 320, 3, 1, 1, 0), (2, 16, 12, 320, 64, 3, 2, 1, 0), (2, 8, 8, 128, 160, 3, 1, 1, 1),
    (2, 8, 8, 8, 320, 3, 1, 1, 0), (1, 32, 32, 320, 8, 3, 1, 1, 0), (5, 8, 16, 192, 128, 1, 1, 0, 0),
    (2, 7, 9, 64, 64, 3, 1, 1, 0)])
def test_conv(ops, nb, h, w, cin, cout, kk, stride, pad, ups):
    x = rnd(nb, h, w, cin)
    wt = rnd(cout, cin, kk, kk, scale=(cin * kk * kk) ** -0.5, seed=1)
    bias = rnd(cout, seed=2, dtype=torch.float32)
    w2d = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    g = ops.ConvGeom(nb, h, w, kk, kk, stride, pad, ups)
    out = ops.gemm(x.view(nb * h * w, cin), w2d, bias, geom=g)
    ref = _conv_ref(x, wt, bias, stride, pad, ups)
    assert (g.h_out, g.w_out) == tuple(ref.shape[1:3])
    check(out.view(nb, g.h_out, g.w_out, cout), ref, f"conv k{kk} s{stride} ups{ups} {cin}->{cout}")


def test_conv_dual_source_concat(ops):
    nb, h, w, c1, c2, cout = 2, 8, 8, 128, 64, 160
    x1, x2 = rnd(nb, h, w, c1), rnd(nb, h, w, c2, seed=7)
    wt = rnd(cout, c1 + c2, 3, 3, scale=(9 * (c1 + c2)) ** -0.5, seed=1)
    bias = rnd(cout, seed=2, dtype=torch.float32)
    w2d = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    g = ops.ConvGeom(nb, h, w, 3, 3, 1, 1)
    out = ops.gemm(x1.view(-1, c1), w2d, bias, geom=g, a2=x2.view(-1, c2))
    check(out.view(nb, h, w, cout), _conv_ref(torch.cat([x1, x2], -1), wt, bias, 1, 1, 0), "conv3x3 concat sources")
    w1 = rnd(cout, c1 + c2, scale=(c1 + c2) ** -0.5, seed=3)
    out = ops.gemm(x1.view(-1, c1), w1, bias, a2=x2.view(-1, c2))
    check(out, torch.cat([x1, x2], -1).view(-1, c1 + c2).float() @ w1.float().t() + bias, "1x1 concat shortcut")


@pytest.mark.parametrize("m,c", [(300, 64), (256, 320), (130, 1280)])
def test_geglu(ops, m, c):
    from v_express_amd import weights as Wt
    a = rnd(m, c)
    w = rnd(8 * c, c, scale=c ** -0.5, seed=1)
    b = rnd(8 * c, seed=2, dtype=torch.float32)
    out = ops.geglu(a, Wt.geglu_interleave(w), Wt.geglu_interleave(b))
    hg = a.float() @ w.float().t() + b
    hval, gate = hg.chunk(2, dim=-1)
    check(out, hval * F.gelu(gate), f"geglu c={c}")


@pytest.mark.parametrize("seqs,n_tok,c,heads", [(3, 64, 64, 8), (2, 256, 320, 8), (4, 16, 128, 8), (6, 4, 64, 8),
                                                (5, 1, 64, 8), (2, 128, 512, 1)])
def test_gemm_split_qkv_vt(ops, seqs, n_tok, c, heads):
    m, d = seqs * n_tok, c // heads
    a = rnd(m, c)
    w = rnd(3 * c, c, scale=c ** -0.5, seed=1)
    bias = rnd(3 * c, seed=2, dtype=torch.float32)
    q = torch.zeros(m, c, device="cuda", dtype=BF)
    k = torch.zeros(m, c, device="cuda", dtype=BF)
    vt = ops.alloc_vt(seqs, heads, d, n_tok, "cuda")
    ops.gemm_split(a, w, bias, [("rows", q), ("rows", k), ("vt", vt)], part_cols=c, seq_len=n_tok, head_dim=d)
    ref = a.float() @ w.float().t() + bias
    check(q, ref[:, :c], "split Q")
    check(k, ref[:, c:2 * c], "split K")
    vref = ref[:, 2 * c:].view(seqs, n_tok, heads, d).permute(0, 2, 3, 1)
    check(vt[..., :n_tok], vref, f"split V^T seq_len={n_tok}")
    assert (vt[..., n_tok:] == 0).all()


# ----------------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("frames,hw,c1,c2,groups,silu", [(4, 64, 320, 0, 32, True), (3, 256, 64, 0, 32, False),
                                                         (2, 64, 1280, 640, 32, True), (2, 1024, 128, 0, 32, True),
                                                         (5, 16, 2560, 0, 32, True), (3, 1, 1280, 1280, 32, True),
                                                         (2, 4096, 320, 0, 32, False)])
def test_groupnorm(ops, frames, hw, c1, c2, groups, silu):
    x1 = rnd(frames, hw, c1, scale=2.0) + 0.7
    x1 = x1.to(BF)
    x2 = rnd(frames, hw, c2, seed=5) if c2 else None
    c = c1 + c2
    gamma, beta = rnd(c, seed=1, dtype=torch.float32) * 0.1 + 1, rnd(c, seed=2, dtype=torch.float32) * 0.1
    out = ops.groupnorm(x1, gamma, beta, frames=frames, hw=hw, groups=groups, eps=1e-5, silu=silu, x2=x2)
    x = x1 if x2 is None else torch.cat([x1, x2], -1)
    ref = F.group_norm(x.float().permute(0, 2, 1), groups, gamma, beta, 1e-5).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    check(out, ref, f"groupnorm C={c} hw={hw}", rel=8e-3, mx=2 ** -6)


@pytest.mark.parametrize("frames,H,W,c1,c2,cout", [(3, 8, 8, 320, 0, 320), (2, 16, 8, 128, 64, 160), (2, 6, 12, 64, 0, 64),
                                                   (1, 64, 64, 320, 320, 320)])
def test_groupnorm_padded_output_feeds_pad0_conv(ops, frames, H, W, c1, c2, cout):
    """GroupNorm(+SiLU) written into the zero-bordered (H+2)x(W+2) image + pad-0 3x3 conv (vx_gemm fast addressing)
    == GroupNorm + pad-1 conv; the border stays zero and the interior equals the plain output bit for bit."""
    hw, c = H * W, c1 + c2
    x1 = (rnd(frames, hw, c1, scale=2.0) + 0.3).to(BF)
    x2 = rnd(frames, hw, c2, seed=5) if c2 else None
    gamma, beta = rnd(c, seed=1, dtype=torch.float32) * 0.1 + 1, rnd(c, seed=2, dtype=torch.float32) * 0.1
    plain = ops.groupnorm(x1, gamma, beta, frames=frames, hw=hw, groups=32, eps=1e-5, silu=True, x2=x2)
    for _ in range(2):   # twice: the persistent buffer is reused
        padded = ops.groupnorm(x1, gamma, beta, frames=frames, hw=hw, groups=32, eps=1e-5, silu=True, x2=x2,
                               pad_hw=(H, W))
    img = padded.view(frames, H + 2, W + 2, c)
    assert torch.equal(img[:, 1:-1, 1:-1], plain.view(frames, H, W, c))
    assert (img[:, 0] == 0).all() and (img[:, -1] == 0).all() and (img[:, :, 0] == 0).all() and (img[:, :, -1] == 0).all()
    wt = rnd(cout, c, 3, 3, scale=(9 * c) ** -0.5, seed=3)
    bias = rnd(cout, seed=4, dtype=torch.float32)
    w2d = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    out = ops.gemm(padded.view(frames * (H + 2) * (W + 2), c), w2d, bias, geom=ops.ConvGeom(frames, H + 2, W + 2, 3, 3, 1, 0))
    ref = _conv_ref(plain.view(frames, H, W, c), wt, bias, 1, 1, 0)
    check(out.view(frames, H, W, cout), ref, f"padded GN -> pad-0 conv {c}->{cout}")
    same = ops.gemm(plain.view(frames * hw, c), w2d, bias, geom=ops.ConvGeom(frames, H, W, 3, 3, 1, 1))
    assert torch.equal(out, same), "fast (pre-padded) and general conv paths must agree bit for bit"


@pytest.mark.parametrize("rows,c", [(100, 64), (257, 320), (64, 640), (33, 1280)])
def test_layernorm(ops, rows, c):
    x = (rnd(rows, c, scale=1.5) + 0.3).to(BF)
    gamma, beta = rnd(c, seed=1, dtype=torch.float32) * 0.1 + 1, rnd(c, seed=2, dtype=torch.float32) * 0.1
    out = ops.layernorm(x, gamma, beta)
    check(out, F.layer_norm(x.float(), (c,), gamma, beta, 1e-5), f"layernorm c={c}")


def test_layernorm_add_table_and_strided(ops):
    b, f, hw, c = 2, 4, 8, 320
    x = rnd(b * f * hw, 2 * c)[:, c:]
    gamma, beta = rnd(c, seed=1, dtype=torch.float32) * 0.1 + 1, rnd(c, seed=2, dtype=torch.float32) * 0.1
    pe = rnd(32, c, seed=3, dtype=torch.float32)
    out = ops.layernorm(x, gamma, beta, add=pe, add_rows_per_entry=hw, add_entries=f)
    ref = F.layer_norm(x.float(), (c,), gamma, beta, 1e-5).view(b, f, hw, c) + pe[:f].view(1, f, 1, c)
    check(out, ref.reshape(-1, c), "layernorm + positional table")


# ----------------------------------------------------------------------------------------------------- attention
def _sdpa_ref(q, k, v):
    return F.scaled_dot_product_attention(q.float(), k.float(), v.float())


@pytest.mark.parametrize("batch,heads,n_q,n_kv,d", [(2, 8, 64, 64, 8), (2, 8, 256, 256, 40), (3, 8, 100, 100, 16),
                                                    (2, 8, 64, 192, 80), (2, 8, 128, 128, 160), (1, 8, 1024, 1024, 40),
                                                    (2, 8, 16, 16, 32), (3, 8, 4, 4, 160), (2, 8, 1, 1, 80),
                                                    (1, 1, 320, 320, 512), (2, 4, 72, 200, 64)])
def test_flash_attention(ops, batch, heads, n_q, n_kv, d):
    c = heads * d
    q = rnd(batch * n_q, c)
    k = rnd(batch * n_kv, c, seed=1)
    v = rnd(batch * n_kv, c, seed=2)
    vt = ops.alloc_vt(batch, heads, d, n_kv, "cuda")
    vt[..., :n_kv] = v.view(batch, n_kv, heads, d).permute(0, 2, 3, 1)
    out = ops.attention(q, k, vt, batch=batch, heads=heads, n_q=n_q, n_kv=n_kv, head_dim=d)
    ref = _sdpa_ref(q.view(batch, n_q, heads, d).transpose(1, 2), k.view(batch, n_kv, heads, d).transpose(1, 2),
                    v.view(batch, n_kv, heads, d).transpose(1, 2)).transpose(1, 2).reshape(batch * n_q, c)
    check(out, ref, f"attention d={d} nq={n_q} nkv={n_kv}", rel=1e-2, mx=2 ** -6)


def test_flash_attention_shared_kv_and_strided_q(ops):
    """reference attention: the f frames of a batch row share one bank (q_per_kv = f); Q is a column slice."""
    f, heads, n, d = 4, 8, 64, 40
    c = heads * d
    qkv = rnd(f * n, 3 * c)
    k, v = rnd(n, c, seed=1), rnd(n, c, seed=2)
    vt = ops.alloc_vt(1, heads, d, n, "cuda")
    vt[..., :n] = v.view(1, n, heads, d).permute(0, 2, 3, 1)
    out = ops.attention(qkv[:, c:2 * c], k, vt, batch=f, heads=heads, n_q=n, n_kv=n, head_dim=d, q_per_kv=f)
    q4 = qkv[:, c:2 * c].reshape(f, n, heads, d).transpose(1, 2)
    k4 = k.view(1, n, heads, d).transpose(1, 2).expand(f, -1, -1, -1)
    v4 = v.view(1, n, heads, d).transpose(1, 2).expand(f, -1, -1, -1)
    check(out, _sdpa_ref(q4, k4, v4).transpose(1, 2).reshape(f * n, c), "attention shared K/V", rel=1e-2, mx=2 ** -6)


def test_flash_attention_softmax_rescale_path(ops):
    """A key spike late in the sequence forces the running-max rescale of the accumulator."""
    heads, n, d = 8, 256, 40
    c = heads * d
    q, k, v = rnd(n, c), rnd(n, c, seed=1), rnd(n, c, seed=2)
    k = k.float()
    k[200] = q[5].float() * 6.0
    k = k.to(BF)
    vt = ops.alloc_vt(1, heads, d, n, "cuda")
    vt[..., :n] = v.view(1, n, heads, d).permute(0, 2, 3, 1)
    out = ops.attention(q, k, vt, batch=1, heads=heads, n_q=n, n_kv=n, head_dim=d)
    ref = _sdpa_ref(q.view(1, n, heads, d).transpose(1, 2), k.view(1, n, heads, d).transpose(1, 2),
                    v.view(1, n, heads, d).transpose(1, 2)).transpose(1, 2).reshape(n, c)
    check(out, ref, "attention with late max spike", rel=1e-2, mx=2 ** -6)


def test_flash_attention_prescaled_keys(ops):
    """`k_prescaled` (vx_attention scale = 0): K carries d^-1/2 log2(e), folded into the to_k weights by
    weights.key_fold so that neither operand is rounded twice; the kernels take q . k as the base-2 logit.  d = 40 runs
    attn3's unit-scale body, d = 80 the plain kernel.  Reference: fp32 SDPA on the keys divided by the fold."""
    for heads, n, d, qs in ((8, 300, 40, 1.0), (8, 300, 40, 8.0), (8, 256, 80, 1.0)):
        c = heads * d
        fold = d ** -0.5 * 1.4426950408889634
        q = (rnd(2 * n, c).float() * qs).to(BF)
        kt = (rnd(2 * n, c, seed=1).float() * qs * fold).to(BF)          # what the key projection would emit
        v = rnd(2 * n, c, seed=2)
        vt = ops.alloc_vt(2, heads, d, n, "cuda")
        vt[..., :n] = v.view(2, n, heads, d).permute(0, 2, 3, 1)
        out = ops.attention(q, kt, vt, batch=2, heads=heads, n_q=n, n_kv=n, head_dim=d, k_prescaled=True)
        k_eff = kt.float() / fold
        ref = _sdpa_ref(q.view(2, n, heads, d).transpose(1, 2), k_eff.view(2, n, heads, d).transpose(1, 2),
                        v.view(2, n, heads, d).transpose(1, 2)).transpose(1, 2).reshape(2 * n, c)
        check(out, ref, f"attention with prescaled keys d={d} qs={qs}", rel=1e-2, mx=2 ** -6)


def test_key_norm_max(ops):
    # (8 heads and >= 512 keys: the row-coalesced kernel with the atomic maximum, incl. a ragged last block and a table that
    # held larger values before the call; everything else: one block per (batch, head))
    for kvb, heads, n, d in ((3, 8, 100, 40), (1, 8, 4096, 40), (2, 4, 7, 64), (5, 8, 1000, 40), (2, 8, 1024, 80), (3, 8, 513, 160)):
        k = rnd(kvb * n, heads * d, seed=n)
        got = ops.key_norm_max(k, kv_batches=kvb, heads=heads, n_kv=n, head_dim=d)
        ref = k.float().view(kvb, n, heads, d).norm(dim=-1).amax(dim=1).reshape(-1)
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6), (got, ref)
        assert torch.equal(got, ops.key_norm_max(k, kv_batches=kvb, heads=heads, n_kv=n, head_dim=d))


@pytest.mark.parametrize("qs,ks,spike", [(1.0, 1.0, 0.0), (6.0, 6.0, 0.0), (25.0, 25.0, 0.0), (1.0, 1.0, 300.0),
                                         (3.0, 3.0, 40.0)])
def test_bounded_softmax_attention_matches_exact(ops, qs, ks, spike):
    """vx_attention_bounded (d = 40): the Cauchy-Schwarz shift c |q_i| Kmax instead of the running row max.
    qs / ks scale the query / key magnitudes (scores up to ~ +-qs*ks*40/sqrt(40)); `spike` plants one key with a huge
    norm that is orthogonal-ish to most queries: Kmax becomes useless as a bound for them, every probability underflows
    under the shift, and the in-kernel check must send those blocks through the exact recompute.  In all cases the
    result has to agree with the exact kernel (vx_attention) and with the fp32 reference."""
    batch, heads, n_q, n_kv, d = 2, 8, 200, 333, 40
    c = heads * d
    q = (rnd(batch * n_q, c).float() * qs).to(BF)
    k = rnd(batch * n_kv, c, seed=1).float() * ks
    if spike:
        k[17] = spike * torch.sign(k[17])                       # |k| = spike * sqrt(40) per head
        k[n_kv + 250] = -spike * torch.sign(k[n_kv + 250])
    k = k.to(BF)
    v = rnd(batch * n_kv, c, seed=2)
    vt = ops.alloc_vt(batch, heads, d, n_kv, "cuda")
    vt[..., :n_kv] = v.view(batch, n_kv, heads, d).permute(0, 2, 3, 1)
    kw = dict(batch=batch, heads=heads, n_q=n_q, n_kv=n_kv, head_dim=d)
    assert ops._BOUNDED_SOFTMAX[0]
    bounded = ops.attention(q, k, vt, **kw)
    pre = ops.attention(q, k, vt, kmax=ops.key_norm_max(k, kv_batches=batch, heads=heads, n_kv=n_kv, head_dim=d), **kw)
    assert torch.equal(bounded, pre)
    try:
        ops._BOUNDED_SOFTMAX[0] = False
        exact = ops.attention(q, k, vt, **kw)
    finally:
        ops._BOUNDED_SOFTMAX[0] = True
    ref = _sdpa_ref(q.view(batch, n_q, heads, d).transpose(1, 2), k.view(batch, n_kv, heads, d).transpose(1, 2),
                    v.view(batch, n_kv, heads, d).transpose(1, 2)).transpose(1, 2).reshape(batch * n_q, c)
    assert torch.isfinite(bounded).all()
    check(exact, ref, f"exact attention qs={qs} spike={spike}", rel=1e-2, mx=2 ** -6)
    check(bounded, ref, f"bounded attention qs={qs} spike={spike}", rel=1e-2, mx=2 ** -6)
    check(bounded, exact, f"bounded vs exact qs={qs} spike={spike}", rel=6e-3, mx=2 ** -6)


@pytest.mark.parametrize("b,f,hw,heads,d", [(2, 4, 16, 8, 8), (2, 16, 64, 8, 40), (1, 24, 16, 8, 80), (2, 8, 4, 8, 160),
                                            (1, 1, 8, 8, 16), (2, 32, 5, 8, 32)])
def test_temporal_attention(ops, b, f, hw, heads, d):
    c = heads * d
    qkv = rnd(b * f * hw, 3 * c)
    out = ops.temporal_attention(qkv, b=b, f=f, hw=hw, heads=heads, head_dim=d)
    t = qkv.view(b, f, hw, 3, heads, d).permute(3, 0, 2, 4, 1, 5)        # [3, b, hw, heads, f, d]
    ref = _sdpa_ref(t[0], t[1], t[2]).permute(0, 3, 1, 2, 4).reshape(b * f * hw, c)
    check(out, ref, f"temporal attention f={f} d={d}", rel=1e-2, mx=2 ** -6)


@pytest.mark.parametrize("batch,n_q,n_kv,heads,d", [(4, 64, 5, 8, 40), (3, 100, 1, 8, 8), (2, 16, 5, 8, 160),
                                                    (2, 7, 16, 8, 16)])
def test_small_kv_attention(ops, batch, n_q, n_kv, heads, d):
    c = heads * d
    q = rnd(batch * n_q, c)
    kv = rnd(batch * n_kv, 2 * c, seed=1)
    out = ops.small_kv_attention(q, kv, batch=batch, n_q=n_q, n_kv=n_kv, heads=heads, head_dim=d)
    k, v = kv[:, :c], kv[:, c:]
    ref = _sdpa_ref(q.view(batch, n_q, heads, d).transpose(1, 2), k.reshape(batch, n_kv, heads, d).transpose(1, 2),
                    v.reshape(batch, n_kv, heads, d).transpose(1, 2)).transpose(1, 2).reshape(batch * n_q, c)
    check(out, ref, f"small-kv attention n_kv={n_kv} d={d}", rel=1e-2, mx=2 ** -6)


@pytest.mark.parametrize("frames,H,W,cin,cout,items", [(4, 32, 32, 640, 640, 2), (2, 16, 16, 1280, 1280, 2), (3, 16, 24, 128, 64, 3),
                                                       (2, 64, 64, 512, 512, 2)])
def test_upsample_conv_as_four_phase_convolutions(ops, frames, H, W, cin, cout, items):
    """weights.fold_upsample_phases + ops.upsample_conv_phases (nearest-2x upsampling + conv3x3 as four 2x2 convolutions over
    the original image: Upsample3D, modules/resnet.py:53-90) against F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"))
    in fp32 on the same rounded inputs; and the four launches of a subset of the items have the bits of the batched call."""
    from v_express_amd import weights
    g_ = torch.Generator().manual_seed(cin + H)
    wt = torch.randn(cout, cin, 3, 3, generator=g_) * (9 * cin) ** -0.5
    bias = torch.randn(cout, generator=g_) * 0.1
    Pw = weights.fold_upsample_phases({"u.weight": wt, "u.bias": bias}, "u", "cuda")
    assert tuple(Pw.w.shape) == (4, cout, 4 * cin)
    x = rnd(frames, H * W, cin)
    with ops.frame_rows(H * W, items=items):
        out = ops.upsample_conv_phases(x, Pw.w, Pw.b, frames=frames, H=H, W=W)
    xi = x.float().view(frames, H, W, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(F.interpolate(xi, scale_factor=2.0, mode="nearest"), wt.to(BF).float().cuda(), bias.cuda(), padding=1)
    check(out, ref.permute(0, 2, 3, 1).reshape(frames, 4 * H * W, cout), f"upsample conv phases {H}x{W} {cin}->{cout}", rel=8e-3, mx=2 ** -6)
    per = frames // items
    if per and frames % items == 0 and items > 1:
        with ops.frame_rows(H * W, items=1):
            one = ops.upsample_conv_phases(x[:per].contiguous(), Pw.w, Pw.b, frames=per, H=H, W=W)
        assert torch.equal(one, out[:per])


@pytest.mark.parametrize("m,c,hw", [(8192, 640, 1024), (2048, 1280, 256)])
def test_ff_proj_fold_dual_source_gemm(ops, m, c, hw):
    """weights.fold_ff_proj: a transformer block's proj_out folded into the feed-forward's second linear -
    out = [h | g] [Wp | Wp W2]^T + (bp + Wp b2) + x_in as ONE dual-source GEMM (K = 5C) - against the fp32 statement of the
    two layers it replaces (modules/transformer_3d.py:150-169: ff.net.2 + residual, then proj_out + residual)."""
    from v_express_amd import weights
    g_ = torch.Generator().manual_seed(c)
    sd = {"ff.weight": torch.randn(c, 4 * c, generator=g_) * (4 * c) ** -0.5, "ff.bias": torch.randn(c, generator=g_) * 0.1,
          "po.weight": (torch.randn(c, c, generator=g_) * c ** -0.5).view(c, c, 1, 1), "po.bias": torch.randn(c, generator=g_) * 0.1}
    Fp = weights.fold_ff_proj(sd, "ff", "po", "cuda")
    assert tuple(Fp.w.shape) == (c, 5 * c) and Fp.w.dtype == BF
    h, gg, x_in = rnd(m, c), rnd(m, 4 * c, seed=1), rnd(m, c, seed=2)
    with ops.frame_rows(hw, items=2):
        out = ops.gemm(h, Fp.w, Fp.b, a2=gg, residual=x_in, gn=(32, hw))
    w2, b2 = sd["ff.weight"].cuda(), sd["ff.bias"].cuda()
    wp, bp = sd["po.weight"].view(c, c).cuda(), sd["po.bias"].cuda()
    h1 = h.float() + gg.float() @ w2.t() + b2
    ref = x_in.float() + h1 @ wp.t() + bp
    check(out, ref, f"ff_proj fold c={c}", rel=8e-3, mx=2 ** -6)
    assert ops.gn_of(out) is not None            # the next GroupNorm's partial sums ride in the same launch
    # frame shards (blocks._motion_module): the same GEMM into float32 (accumulator + bias), the residual added afterwards by
    # vx_add_residual_f32 - the bits of the one-launch form, whichever tile kernels the two launches take
    with ops.frame_rows(hw, items=2):
        y32 = ops.gemm(h, Fp.w, Fp.b, a2=gg, out_f32=True)
    assert torch.equal(ops.add_residual_f32(x_in, y32), out)


@pytest.mark.parametrize("frames,hw,c,alpha", [(3, 64, 1280, 3.0), (2, 256, 1280, 3.0), (4, 1024, 640, 3.0), (16, 4096, 320, 3.0),
                                               (1, 48, 320, 0.5)])
def test_audio_xattn_one_launch(ops, frames, hw, c, alpha):
    """vx_audio_xattn + vx_audio_xattn_pack (the audio cross-attention of a spatial transformer block as two 48-column products
    with per-frame operands) against the fp32 statement of the block it replaces: modules/mutual_self_attention.py:227-244 ->
    h + w * to_out(softmax(to_q(LN(h)) k^T / sqrt d) v), 8 heads, 5 audio tokens per frame.  Both statistics formats ([m, 2]
    and the two-part [m, 4] of the 32x32 level), in place, the statistics of the written rows, 16- and 32-row waves; and a
    launch over a subset of the frames is bit-identical to those rows of the full launch (batch invariance)."""
    from v_express_amd import weights
    heads, n_ctx, d = 8, 5, c // 8
    m = frames * hw
    assert ops.audio_xattn_applies(c, heads, n_ctx, hw)
    g = torch.Generator().manual_seed(c + hw)
    h = (torch.randn(m, c, generator=g) * 1.5 + 0.3).cuda().to(BF)
    kv = (torch.randn(frames * n_ctx, 2 * c, generator=g) * 1.2).cuda().to(BF)
    wq = torch.randn(c, c, generator=g) * c ** -0.5
    wo = (torch.randn(c, c, generator=g) * c ** -0.5).cuda().to(BF)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    bo = (torch.randn(c, generator=g) * 0.3).cuda()
    Fq = weights.fold_layernorm(wq, None, gamma, beta, "cuda")
    two = c == 640
    st = torch.empty((m, 4 if two else 2), device="cuda", dtype=torch.float32)
    ops.row_stats(h, 1e-5, out=st)
    fold = ops.audio_xattn_pack(kv, Fq.w, Fq.b, wo, frames=frames, n_ctx=n_ctx, heads=heads)
    st_out = torch.empty_like(st)
    out = torch.empty_like(h)
    ops.audio_xattn(h, st, fold, bo, alpha, rows_per_frame=hw, stats_out=st_out, out=out)
    # fp32 reference on the same rounded inputs
    x = h.float()
    ln = F.layer_norm(x, (c,), gamma.cuda(), beta.cuda(), 1e-5)
    q = (ln @ wq.cuda().t()).view(frames, hw, heads, d).transpose(1, 2)
    k = kv[:, :c].float().view(frames, n_ctx, heads, d).transpose(1, 2)
    v = kv[:, c:].float().view(frames, n_ctx, heads, d).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1) @ v
    ref = x + alpha * (a.transpose(1, 2).reshape(m, c) @ wo.float().t() + bo)
    check(out, ref, f"audio_xattn c={c} hw={hw}", rel=8e-3, mx=2 ** -6)
    # the block's own contribution, without the residual that dominates the norm above
    check(out.float() - x, ref - x, f"audio_xattn increment c={c} hw={hw}", rel=2.5e-2, mx=2 ** -4)
    # statistics of the STORED rows
    o = out.float()
    if two:
        want = torch.stack([o[:, :c // 2].sum(1), o[:, :c // 2].pow(2).sum(1), o[:, c // 2:].sum(1), o[:, c // 2:].pow(2).sum(1)], 1)
        assert torch.allclose(st_out, want, rtol=2e-4, atol=2e-2), (st_out - want).abs().max()
    else:
        mean, var = o.mean(1), o.var(1, unbiased=False)
        assert torch.allclose(st_out[:, 0], mean, rtol=1e-4, atol=1e-4)
        assert torch.allclose(st_out[:, 1], (var + 1e-5).rsqrt(), rtol=2e-3)
    # in place, statistics written over the ones read
    h2, st2 = h.clone(), st.clone()
    ops.audio_xattn(h2, st2, fold, bo, alpha, rows_per_frame=hw, stats_out=st2)
    assert torch.equal(h2, out) and torch.equal(st2, st_out)
    # a subset of the frames: same bits
    if frames > 1:
        f0 = frames // 2
        sub = ops.audio_xattn_pack(kv[f0 * n_ctx:], Fq.w, Fq.b, wo, frames=frames - f0, n_ctx=n_ctx, heads=heads)
        o2 = torch.empty_like(h[f0 * hw:])
        ops.audio_xattn(h[f0 * hw:], st[f0 * hw:].contiguous(), sub, bo, alpha, rows_per_frame=hw, out=o2)
        assert torch.equal(o2, out[f0 * hw:])


# ----------------------------------------------------------------------------------------------------- elementwise
def test_add_row_bias(ops):
    x = rnd(50, 640)
    bias = rnd(320, seed=1, dtype=torch.float32)
    ref = x.float().clone()
    ref[:, 320:] += 0.95 * bias
    ops.add_row_bias(x[:, 320:], bias, 0.95)
    check(x, ref, "add_row_bias on a strided view")


def test_layout_and_loop_kernels(ops):
    from oracle import loop as OL
    F_, h, w, f = 11, 8, 8, 4
    hw = h * w
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(1, 4, F_, h, w, generator=g).cuda()
    ids = torch.tensor([8, 9, 10, 9], dtype=torch.int32, device="cuda")
    x = ops.gather_latents(lat, ids, reps=2)
    ref = lat[0][:, ids.long()].permute(1, 2, 3, 0).reshape(f, hw, 4)
    assert x.shape == (2 * f, hw, 8)
    check(x[:f, :, :4], ref, "gather_latents", rel=4e-3, mx=2 ** -8)
    assert torch.equal(x[:f], x[f:]) and (x[..., 4:] == 0).all()
    # NCFHW <-> NHWC
    t = torch.randn(2, 320, 3, h, w, generator=g).cuda()
    nh = ops.ncfhw_to_nhwc(t, 320)
    check(nh, t.permute(0, 2, 3, 4, 1).reshape(6, hw, 320), "ncfhw_to_nhwc", rel=4e-3, mx=2 ** -8)
    back = ops.nhwc_to_ncfhw(nh.float().view(-1, 320), 2, 320, 3, h, w)
    assert torch.equal(back, nh.float().view(2, 3, h, w, 320).permute(0, 4, 1, 2, 3))
    # CFG combine + overlap/DDIM step against the oracle's loop arithmetic
    windows = OL.uniform_windows(F_, f, 2)
    from v_express_amd.context import overlap_plan
    plan = overlap_plan(windows, F_)
    outs = [torch.randn(2 * f * hw, 8, generator=g).cuda() for _ in windows]
    preds = torch.empty(len(windows), 4, f, hw, device="cuda")
    for wi, o in enumerate(outs):
        ops.cfg_combine(o, 4, f, hw, 3.5, preds[wi])
    ddim = OL.DDIM()
    ddim.set_timesteps(25)
    t_step = 519

    def fake_unet(inp, t, e, kfeat):       # returns the stored window outputs in call order
        o = outs[fake_unet.i].cpu()
        fake_unet.i += 1
        return o.view(2, f, hw, 8)[..., :4].permute(0, 3, 1, 2).reshape(2, 4, f, h, w)
    fake_unet.i = 0
    ref_lat = OL.mean_overlap(fake_unet, lat.cpu(), [t_step], ddim, windows, 3.5,
                              torch.zeros(2, 1, F_, h, w), torch.zeros(2, F_, 1, 8))
    sf = plan["step_frames"]
    terms = torch.full((len(sf), plan["max_terms"], 2), -1, dtype=torch.int32)
    for i, fr in enumerate(sf):
        for j, (wi, li) in enumerate(plan["terms"][fr]):
            terms[i, j, 0], terms[i, j, 1] = wi, li
    from v_express_amd.scheduler import DDIMScheduler
    sch = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                        steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True,
                        timestep_spacing="trailing")
    sch.set_timesteps(25)
    lat2 = lat.clone()
    ops.overlap_ddim_step(lat2, preds, terms.cuda(), torch.tensor(sf, dtype=torch.int32, device="cuda"),
                          torch.tensor([float(plan["counts"][fr]) for fr in sf], device="cuda"),
                          sch.step_coefficients(t_step))
    check(lat2.cpu(), ref_lat, "cfg_combine + overlap_ddim_step (reflected window)", rel=1e-5, mx=1e-5)
    # VAE post-process
    img = torch.randn(2 * hw, 8, generator=g).cuda() * 2
    pp = ops.vae_postprocess(img, 2, 3, h, w)
    assert torch.allclose(pp, (img.view(2, h, w, 8)[..., :3].permute(0, 3, 1, 2) / 2 + 0.5).clamp(0, 1))


def test_errors_are_reported_not_fatal(ops):
    from v_express_amd.lib import VxError
    with pytest.raises(VxError):
        ops.gemm(rnd(16, 12), rnd(8, 12))            # K not a multiple of 8
    with pytest.raises(VxError):
        ops.temporal_attention(rnd(33 * 4, 3 * 64), b=1, f=33, hw=4, heads=8, head_dim=8)


# ------------------------------------------------------------------------------------- fp8 projections (config 5)


def _deq(q8, scale):
    return q8.view(torch.float8_e4m3fn).float() * scale[:, None]


@pytest.mark.parametrize("rows,c,norm,pe", [(300, 320, True, False), (257, 640, True, True), (64, 1280, False, False),
                                            (130, 40, True, False)])
def test_layernorm_fp8(ops, rows, c, norm, pe):
    """vx_layernorm_fp8: y = LN(x) (+ table) or x itself, scale[r] = max|y[r]| / 448, q = e4m3(y / scale), K zero-padded
    to a multiple of 128.  The dequantised row must sit within half an e4m3 ulp of y (3 mantissa bits: 2^-4 relative for
    normal values, 2^-9 * scale... absolute below 2^-6) and the scale must be the row maximum."""
    x = rnd(rows, c, scale=2.0, seed=7)
    g = b = add = None
    y = x.float()
    if norm:
        g, b = 1 + 0.1 * rnd(c, seed=1, dtype=torch.float32), 0.1 * rnd(c, seed=2, dtype=torch.float32)
        y = F.layer_norm(y, (c,), g, b, 1e-5)
        if pe:
            add = rnd(8, c, seed=3, dtype=torch.float32)
            y = y + add[(torch.arange(rows, device="cuda") // 16) % 8]
    r = ops.layernorm_fp8(x, g, b, add=add, add_rows_per_entry=16, add_entries=8) if norm else ops.quantize_fp8(x)
    kp = (c + 127) // 128 * 128
    assert r.q.shape == (rows, kp) and r.q.dtype == torch.uint8 and r.k == c
    assert (r.q[:, c:] == 0).all(), "K padding must be zero bytes"
    amax = y.abs().amax(dim=1)
    assert torch.allclose(r.scale, amax / 448.0, rtol=2e-3), (r.scale[:4], amax[:4] / 448)
    deq = _deq(r.q[:, :c], r.scale)
    err = (deq - y).abs()
    tol = y.abs() * 2 ** -4 + r.scale[:, None] * 2 ** -6 * 1.01 + 2e-3 * amax[:, None]   # half ulp (+ bf16-level LN noise)
    assert (err <= tol).all(), (err.max().item(), (err / tol).max().item())
    assert ((deq - y).pow(2).sum().sqrt() / y.pow(2).sum().sqrt()).item() <= 4e-2


@pytest.mark.parametrize("m,n,k", [(65536, 320, 320), (384, 960, 640), (200, 1280, 1280), (4096, 320, 320),
                                   (256 * 48, 1920, 640)])      # first / last: the 256x320 tile (>= 256 tiles)
def test_gemm_fp8_store_and_split(ops, m, n, k):
    """The fp8 GEMM against a float64 product of the DEQUANTISED operands (e4m3 x e4m3 products are exact in fp32, so
    only the accumulation order and the bf16 output rounding remain): per-kernel tolerance of the bf16 GEMM."""
    a = rnd(m, k, seed=1)
    w = rnd(n, k, scale=k ** -0.5, seed=2)
    a8, w8 = ops.quantize_fp8(a), ops.fp8_weight(w)
    A = _deq(a8.q, a8.scale).double()
    W = _deq(w8.w8, w8.scale).double()
    bias = rnd(n, seed=3, dtype=torch.float32)
    res = rnd(m, n, seed=4)
    ref = A @ W.t() + bias.double()
    out = ops.gemm(a8, w8, bias, residual=res, alpha=0.95)
    check(out, (res.double() + 0.95 * ref).float(), f"fp8 gemm store {m}x{n}x{k}")
    # the quantisation itself: against the un-quantised bf16 product, e4m3 operand noise (2^-4 / sqrt(12) per element,
    # both operands, averaged over K terms)
    full = a.double() @ w.double().t() + bias.double()
    rl2 = ((ref - full).norm() / full.norm()).item()
    assert rl2 <= 4e-2, rl2
    if n % 3 == 0 and (n // 3) % 16 == 0:
        c = n // 3
        heads, seq = 8, 64 if m % 64 == 0 else m
        if m % seq == 0 and c % heads == 0:
            d = c // heads
            q = torch.empty((m, c), device="cuda", dtype=BF)
            kk = torch.empty((m, c), device="cuda", dtype=BF)
            vt = ops.alloc_vt(m // seq, heads, d, seq, "cuda")
            ops.gemm_split(a8, w8, bias, [("rows", q), ("rows", kk), ("vt", vt)], part_cols=c, seq_len=seq, head_dim=d)
            check(q, ref[:, :c].float(), "fp8 split q")
            check(kk, ref[:, c:2 * c].float(), "fp8 split k")
            v_ref = ref[:, 2 * c:].float().view(m // seq, seq, heads, d).permute(0, 2, 3, 1)
            check(vt[..., :seq], v_ref, "fp8 split v^T")


def test_gemm_fp8_ring_vs_classic_tiles(ops):
    """The persistent ring kernel's fp8 instantiation (opt-in: it spills and measures slower than the classic fp8 tiles;
    requested per call here - ops.FP8_RING -> vx_gemm_params.ring_hint = 3, ABI 14 - so that every model-level fp8 test
    runs the product default) against the classic fp8 tiles and the float64 reference of the dequantised operands."""
    from v_express_amd import lib as L
    m, n, k = 256 * 200, 640, 640
    a8, w8 = ops.quantize_fp8(rnd(m, k, seed=1)), ops.fp8_weight(rnd(n, k, scale=k ** -0.5, seed=2))
    bias, res = rnd(n, seed=3, dtype=torch.float32), rnd(m, n, seed=4)
    ref = _deq(a8.q, a8.scale).double() @ _deq(w8.w8, w8.scale).double().t() + bias.double()
    outs = {}
    saved = ops.FP8_RING[0]
    try:
        for mode in (2, 0):
            ops.FP8_RING[0] = bool(mode)
            with ops.GemmProfile() as prof:
                outs[mode] = (ops.gemm(a8, w8, bias), ops.gemm(a8, w8, bias, residual=res, alpha=0.5))
            assert _ring_used(prof) == bool(mode), prof.records[0][3]
    finally:
        ops.FP8_RING[0] = saved
    for mode, (plain, with_res) in outs.items():
        check(plain, ref.float(), f"fp8 gemm ring mode {mode}")
        check(with_res, (res.double() + 0.5 * ref).float(), f"fp8 gemm + residual, ring mode {mode}")
    assert torch.allclose(outs[2][0].float(), outs[0][0].float(), rtol=2e-2, atol=1e-3)


def test_gemm_fp8_rejects_what_it_does_not_support(ops):
    from v_express_amd import lib as L
    a8, w8 = ops.quantize_fp8(rnd(64, 320)), ops.fp8_weight(rnd(320, 320, seed=1))
    with pytest.raises(TypeError):
        ops.gemm(a8, rnd(320, 320, seed=1))                  # bf16 weight with fp8 activations
    with pytest.raises(L.VxError):
        ops.geglu(a8, w8, None)                               # GEGLU epilogue is not built for fp8


# --------------------------------------------------------------------------- LayerNorm folded into the consumer GEMM


def _folded(ops, w, bias, gamma, beta, interleave=False):
    from v_express_amd import weights as Wt
    return Wt.fold_layernorm(w.float(), bias, gamma, beta, "cuda", interleave=interleave)


def test_row_stats(ops):
    for rows, c in ((300, 320), (257, 640), (64, 1280), (33, 40)):
        x = (rnd(rows, c, seed=c).float() * 1.5 + 0.7).to(BF)
        st = ops.row_stats(x, 1e-5)
        xf = x.float()
        assert torch.allclose(st[:, 0], xf.mean(dim=1), rtol=1e-5, atol=1e-6)
        assert torch.allclose(st[:, 1], torch.rsqrt(xf.var(dim=1, unbiased=False) + 1e-5), rtol=1e-5)


@pytest.mark.parametrize("m,n,k,offset", [(256 * 96, 640, 320, 0.0), (256 * 96, 640, 320, 3.0), (300, 320, 640, 0.5),
                                          (2048, 1280, 1280, 1.0), (256 * 200, 320, 320, 0.5)])
def test_gemm_with_folded_layernorm(ops, m, n, k, offset):
    """vx_gemm_params.ln_stats: x W'^T transformed to rstd (acc - mean colsum) + b' must equal LN(x) W^T + b computed in
    fp32 - on the ring kernel (first / last case) and the classic tiles, with a row mean of up to 3 standard deviations
    (the cancellation the algebra introduces), residual + alpha + row-bias epilogue options included."""
    x = (rnd(m, k, seed=1).float() + offset).to(BF)
    w = rnd(n, k, scale=k ** -0.5, seed=2)
    bias = rnd(n, seed=3, dtype=torch.float32)
    gamma, beta = 1 + 0.1 * rnd(k, seed=4, dtype=torch.float32), 0.1 * rnd(k, seed=5, dtype=torch.float32)
    Fd = _folded(ops, w, bias, gamma, beta)
    ref = F.layer_norm(x.float(), (k,), gamma, beta, 1e-5) @ w.float().t() + bias
    st = ops.row_stats(x)
    out = ops.gemm(x, Fd.w, Fd.b, ln=(st, Fd.s))
    check(out, ref, f"folded LN gemm {m}x{n}x{k} offset {offset}", rel=8e-3, mx=2 ** -6)
    res = rnd(m, n, seed=6)
    grp = 256 if m % 256 == 0 else m
    rowbias = rnd(m // grp, n, seed=7, dtype=torch.float32)
    out2 = ops.gemm(x, Fd.w, Fd.b, ln=(st, Fd.s), residual=res, alpha=0.9, rowbias=rowbias, rows_per_group=grp)
    check(out2, res.float() + 0.9 * (ref + rowbias.repeat_interleave(grp, 0)), "folded LN gemm + epilogue options",
          rel=8e-3, mx=2 ** -6)


def test_geglu_and_split_with_folded_layernorm(ops):
    m, c, heads = 256 * 24, 320, 8
    x = (rnd(m, c, seed=1).float() * 1.3 + 0.4).to(BF)
    gamma, beta = 1 + 0.1 * rnd(c, seed=4, dtype=torch.float32), 0.1 * rnd(c, seed=5, dtype=torch.float32)
    ln = F.layer_norm(x.float(), (c,), gamma, beta, 1e-5)
    st = ops.row_stats(x)
    # GEGLU: value rows then gate rows in the source layout; fold_layernorm interleaves them for the kernel
    w1, b1 = rnd(8 * c, c, scale=c ** -0.5, seed=2), rnd(8 * c, seed=3, dtype=torch.float32)
    Fd = _folded(ops, w1, b1, gamma, beta, interleave=True)
    y = ln @ w1.float().t() + b1
    ref = y[:, :4 * c] * F.gelu(y[:, 4 * c:])
    check(ops.geglu(x, Fd.w, Fd.b, ln=(st, Fd.s)), ref, "folded LN geglu", rel=8e-3, mx=2 ** -6)
    # fused QKV with the V^T part
    wq = rnd(3 * c, c, scale=c ** -0.5, seed=6)
    Fq = _folded(ops, wq, None, gamma, beta)
    d, seq = c // heads, 64
    q = torch.empty((m, c), device="cuda", dtype=BF)
    kk = torch.empty((m, c), device="cuda", dtype=BF)
    vt = ops.alloc_vt(m // seq, heads, d, seq, "cuda")
    ops.gemm_split(x, Fq.w, Fq.b, [("rows", q), ("rows", kk), ("vt", vt)], part_cols=c, seq_len=seq, head_dim=d,
                   ln=(st, Fq.s))
    r = ln @ wq.float().t()
    check(q, r[:, :c], "folded LN split q", rel=8e-3, mx=2 ** -6)
    check(kk, r[:, c:2 * c], "folded LN split k", rel=8e-3, mx=2 ** -6)
    check(vt[..., :seq], r[:, 2 * c:].view(m // seq, seq, heads, d).permute(0, 2, 3, 1), "folded LN split v^T", rel=8e-3,
          mx=2 ** -6)


# ------------------------------------------------------------------- round 3: producer-side statistics, GroupNorm fold
@pytest.mark.parametrize("m,n,k,res,offset", [(256 * 384, 320, 320, True, 0.0), (256 * 200, 320, 1280, True, 2.5),
                                              (256 * 192, 320, 64, False, 0.5), (256 * 96, 640, 320, True, 0.5),
                                              (300, 320, 640, True, 1.0), (256 * 192, 320, 320, True, 30.0)])
def test_gemm_row_stats_out(ops, m, n, k, res, offset):
    """vx_gemm_params.row_stats_out: (mean, rstd) of every STORED bf16 output row, against float64 statistics of the
    tensor the launch wrote - from the ring epilogue's registers when one 256 x 320 tile holds whole rows (n = 320; sums
    of x and x^2 in float32, rows with a mean of up to 2.5 standard deviations), from vx_row_stats inside vx_gemm
    otherwise (n = 640 on the ring kernel, the classic tiles).  The output itself must not change.

    Last row: |mean| / std of about 30 - the precision statement of vexpress_hip.h for the one-pass form (E[x^2] - mean^2
    from float32 sums of 320 values cancels ~10 bits there): rstd within 3e-3 instead of 1e-4.  The model's residual
    stream has |mean| / std < 1 in every LayerNorm input (tests/test_gpu_fullsize.py would not hold otherwise)."""
    a, w = rnd(m, k), rnd(n, k, scale=k ** -0.5, seed=1)
    bias = rnd(n, seed=2, dtype=torch.float32) + offset
    r = rnd(m, n, seed=3) if res else None
    st = torch.full((m, 2), float("nan"), device="cuda")
    with ops.GemmProfile() as prof:
        out = ops.gemm(a, w, bias, residual=r, alpha=0.75 if res else 1.0, stats_out=st, stats_eps=1e-5)
    if m % 256 == 0:
        assert _ring_used(prof), prof.records[0][3]
    plain = ops.gemm(a, w, bias, residual=r, alpha=0.75 if res else 1.0)
    assert torch.equal(out, plain)
    x = out.double()
    mean, rstd = x.mean(dim=1), torch.rsqrt(x.var(dim=1, unbiased=False) + 1e-5)
    assert torch.isfinite(st).all()
    assert torch.allclose(st[:, 0].double(), mean, rtol=2e-5, atol=2e-6), (st[:, 0].double() - mean).abs().max()
    rtol = 1e-4 if abs(offset) < 10 else 3e-3
    assert torch.allclose(st[:, 1].double(), rstd, rtol=rtol), ((st[:, 1].double() - rstd) / rstd).abs().max()
    if abs(offset) >= 10:
        sd = x.std(dim=1, unbiased=False)
        assert float((mean.abs() / sd).median()) > 15, "the row is supposed to test a large mean-to-deviation ratio"
    # a launch over the first half of the rows gives the same bits (batch invariance of the fused statistics)
    if (m // 2) % 256 == 0 and (m // 2 // 256) * (n // 320) >= 192:
        st2 = torch.empty((m // 2, 2), device="cuda")
        ops.gemm(a[:m // 2], w, bias, residual=None if r is None else r[:m // 2], alpha=0.75 if res else 1.0,
                 stats_out=st2)
        assert torch.equal(st2, st[:m // 2])


@pytest.mark.parametrize("frames,hw,c,n", [(32, 1024, 320, 320), (4, 9216, 320, 320), (16, 2304, 640, 640)])
def test_groupnorm_folded_into_linear(ops, frames, hw, c, n):
    """GroupNorm without activation folded into the 1x1 / linear layer behind it (vx_groupnorm_stats +
    vx_groupnorm_fold_linear + vx_gemm_params.w_group_rows): against GroupNorm + linear in float32, next to the
    unfused pair of kernels, with per-frame statistics that differ strongly between frames and groups."""
    groups, eps = 32, 1e-6
    g = torch.Generator().manual_seed(5)
    x = torch.randn(frames, hw, c, generator=g)
    x = x * (0.5 + 2.0 * torch.rand(frames, 1, c, generator=g)) + 1.5 * torch.randn(frames, 1, c, generator=g)
    x = x.to("cuda").to(BF)
    w, bias = rnd(n, c, scale=c ** -0.5, seed=1), rnd(n, seed=2, dtype=torch.float32)
    gamma, beta = 1 + 0.2 * rnd(c, seed=3, dtype=torch.float32), 0.3 * rnd(c, seed=4, dtype=torch.float32)
    from v_express_amd import weights as Wt
    G = Wt.fold_groupnorm(w.float(), bias, gamma, beta, "cuda")
    ref = F.group_norm(x.float().transpose(1, 2), groups, gamma, beta, eps).transpose(1, 2).reshape(frames * hw, c)
    ref = ref @ w.float().t() + bias
    with ops.frame_rows(hw, items=1):
        assert ops.gn_fold_applies(frames * hw, hw, c, n)
        ws, sl = ops.groupnorm_stats(x, frames=frames, hw=hw, groups=groups)
        w_f, b_f = ops.groupnorm_fold_linear(ws, G.g, G.w, G.bb, frames=frames, hw=hw, groups=groups, eps=eps, slices=sl)
        st = torch.empty((frames * hw, 2), device="cuda")
        with ops.GemmProfile() as prof:
            out = ops.gemm(x.view(frames * hw, c), w_f, None, rowbias=b_f, rows_per_group=hw, w_group_rows=hw,
                           stats_out=st if n == 320 else None)
        assert _ring_used(prof), prof.records[0][3]
        unfused = ops.gemm(ops.groupnorm(x, gamma, beta, frames=frames, hw=hw, groups=groups, eps=eps,
                                         silu=False).view(frames * hw, c), w, bias)
    check(out, ref, f"GroupNorm folded into linear {frames}x{hw}x{c}->{n}")
    check(unfused, ref, "GroupNorm + linear, unfused")
    # the per-frame weights: one rounding of w * gamma * rstd
    xf = x.float().view(frames, hw, groups, c // groups)
    rstd = torch.rsqrt(xf.var(dim=(1, 3), unbiased=False) + eps).repeat_interleave(c // groups, dim=1)
    check(w_f, w.float()[None] * (gamma[None] * rstd)[:, None, :], "per-frame folded weights", rel=4e-3, mx=2 ** -8)
    if n == 320:
        o = out.double()
        assert torch.allclose(st[:, 0].double(), o.mean(dim=1), rtol=2e-5, atol=2e-6)
    # a lone frame range computed alone is bit-identical (the fold is per frame)
    half = frames // 2
    with ops.frame_rows(hw, items=1):
        if ops.gn_fold_applies(half * hw, hw, c, n):
            ws2, sl2 = ops.groupnorm_stats(x[:half], frames=half, hw=hw, groups=groups)
            w2, b2 = ops.groupnorm_fold_linear(ws2, G.g, G.w, G.bb, frames=half, hw=hw, groups=groups, eps=eps, slices=sl2)
            o2 = ops.gemm(x[:half].reshape(half * hw, c), w2, None, rowbias=b2, rows_per_group=hw, w_group_rows=hw)
            assert torch.equal(o2, out[:half * hw])


def test_grouped_weights_need_the_ring_kernel(ops):
    """w_group_rows on a launch the persistent kernel does not take is refused (VX_ERR_UNSUPPORTED), never silently
    computed with one weight."""
    from v_express_amd.lib import VxError
    x, w = rnd(4 * 64, 64), rnd(4, 64, 64, scale=0.1, seed=1)
    with pytest.raises(VxError):
        ops.gemm(x, w, None, w_group_rows=64)


def _gn_totals(st):
    """(sum, sum of squares) per (frame, group) from a GnStats workspace, in float64."""
    return st.ws.double().sum(dim=1)                     # [frames, groups, 2]


def _gn_check(ops, out, frames, hw, groups, what):
    """The partial sums on `out` against float64 sums of the stored tensor, and the apply pass that consumes them against
    the GroupNorm that runs its own statistics pass."""
    st = ops.gn_of(out)
    assert st is not None and st.fits(frames, hw, groups, out.shape[-1]), f"{what}: no statistics on the output"
    assert torch.isfinite(st.ws).all()
    x = out.double().view(frames, hw, groups, -1)
    want = torch.stack([x.sum(dim=(1, 3)), (x * x).sum(dim=(1, 3))], dim=-1)
    got = _gn_totals(st)
    assert torch.allclose(got, want, rtol=2e-6, atol=1e-3), (what, (got - want).abs().max().item())
    c = out.shape[-1]
    gamma, beta = 1 + 0.2 * rnd(c, seed=3, dtype=torch.float32), 0.3 * rnd(c, seed=4, dtype=torch.float32)
    o3 = ops.keep_gn(out.view(frames, hw, c), out)
    fused = ops.groupnorm(o3, gamma, beta, frames=frames, hw=hw, groups=groups, eps=1e-5, silu=True)
    plain = ops.groupnorm(out.view(frames, hw, c).clone(), gamma, beta, frames=frames, hw=hw, groups=groups, eps=1e-5,
                          silu=True)
    # same apply kernel, statistics equal to ~1e-6: the outputs differ by at most one bf16 rounding here and there
    assert (fused.float() - plain.float()).abs().max().item() <= 2 ** -6 * plain.float().abs().max().item()
    assert (fused != plain).float().mean().item() < 1e-3, (fused != plain).float().mean().item()
    return st


@pytest.mark.parametrize("frames,hw,n,k,res", [(48, 1024, 640, 320, True), (13, 4096, 320, 64, False),
                                               (24, 2304, 320, 128, True), (96, 256, 1280, 128, True)])
def test_gemm_gn_partial_sums_ring_linear(ops, frames, hw, n, k, res):
    """vx_gemm_params.gn_ws on the persistent 256 x 320 kernel (proj_out / the motion module's out-projection,
    modules/transformer_3d.py:154, motion_module.py:177 -> the next block's GroupNorm): per (frame, 128-row slab, group)
    sums of the stored bf16 values; groups of 10 / 20 / 40 channels; output unchanged; half a batch = same bits."""
    m = frames * hw
    a, w = rnd(m, k), rnd(n, k, scale=k ** -0.5, seed=1)
    bias = rnd(n, seed=2, dtype=torch.float32) + 0.7
    r = rnd(m, n, seed=3) if res else None
    with ops.GemmProfile() as prof:
        out = ops.gemm(a, w, bias, residual=r, gn=(32, hw))
    assert _ring_used(prof), prof.records[0][3]
    assert torch.equal(out, ops.gemm(a, w, bias, residual=r))
    st = _gn_check(ops, out, frames, hw, 32, f"ring linear {frames}x{hw}x{n}")
    assert st.slabs == hw // 128
    h = frames // 2
    if (h * hw // 256) * (n // 320) >= 192:
        o2 = ops.gemm(a[:h * hw], w, bias, residual=None if r is None else r[:h * hw], gn=(32, hw))
        assert torch.equal(ops.gn_of(o2).ws, st.ws[:h])


def test_gemm_gn_partial_sums_ring_conv(ops):
    """conv1 of a resnet (modules/resnet.py:223 -> norm2, :235-241): 3x3 over the zero-bordered image with the
    time-embedding rows, through the ring kernel, leaving norm2's partial sums."""
    nb, hh, ww, cin, cout = 48, 32, 32, 64, 320
    x = torch.zeros(nb, hh + 2, ww + 2, cin, device="cuda", dtype=BF)
    x[:, 1:-1, 1:-1] = rnd(nb, hh, ww, cin)
    wt = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=1)
    bias = rnd(cout, seed=2, dtype=torch.float32)
    w2d = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    g = ops.ConvGeom(nb, hh + 2, ww + 2, 3, 3, 1, 0)
    rows = hh * ww * (nb // 2)
    rowbias = rnd(2, cout, seed=5, dtype=torch.float32)
    with ops.GemmProfile() as prof:
        out = ops.gemm(x.view(-1, cin), w2d, bias, geom=g, rowbias=rowbias, rows_per_group=rows, gn=(32, hh * ww))
    assert _ring_used(prof), prof.records[0][3]
    ref = _conv_ref(x, wt, bias, 1, 0, 0).reshape(nb * hh * ww, cout) + rowbias.repeat_interleave(rows, 0)
    check(out, ref, "ring conv with GroupNorm partial sums")
    _gn_check(ops, out, nb, hh * ww, 32, "ring conv")


@pytest.mark.parametrize("frames,hw,n,k,res", [(32, 256, 1280, 1280, True), (6, 256, 640, 192, False),
                                               (32, 64, 1280, 1280, True), (64, 64, 1280, 640, True),
                                               (5, 64, 320, 64, False)])
def test_gemm_gn_partial_sums_classic_tiles(ops, frames, hw, n, k, res):
    """The same on the classic 128 x 160 / 64 x 160 tiles (16x16 and 8x8 levels): 64-row slabs."""
    m = frames * hw
    a, w = rnd(m, k), rnd(n, k, scale=k ** -0.5, seed=1)
    bias = rnd(n, seed=2, dtype=torch.float32) - 0.4
    r = rnd(m, n, seed=3) if res else None
    with ops.frame_rows(hw, items=2 if frames % 2 == 0 else 1), ops.GemmProfile() as prof:
        out = ops.gemm(a, w, bias, residual=r, alpha=0.9, gn=(32, hw))
        base = ops.gemm(a, w, bias, residual=r, alpha=0.9)
    assert not _ring_used(prof), prof.records[0][3]
    assert torch.equal(out, base)
    st = _gn_check(ops, out, frames, hw, 32, f"classic {frames}x{hw}x{n}x{k}")
    assert st.slabs == hw // 64
    # twice the batch (at the 16x16 level that is enough rows for the 256 x 320 tile, which writes no partial sums): the
    # request keeps the launch on the 128 x 160 tile, so the first half of the doubled launch gives the same bits
    a2 = torch.cat([a, rnd(m, k, seed=11)], dim=0)
    r2 = None if r is None else torch.cat([r, rnd(m, n, seed=12)], dim=0)
    with ops.frame_rows(hw, items=4 if frames % 2 == 0 else 2):
        out2 = ops.gemm(a2, w, bias, residual=r2, alpha=0.9, gn=(32, hw))
    st2 = ops.gn_of(out2)
    assert st2 is not None and torch.equal(out2[:m], out) and torch.equal(st2.ws[:frames], st.ws)


def test_gemm_gn_partial_sums_downsample_conv_and_refusals(ops):
    """Downsample3D (modules/resnet.py:106-118: 3x3 stride 2 pad 1, the gather addressing) leaves the next resnet's norm1
    sums; launches that cannot (split-K, fp32 output, groups that straddle a wave's 80 columns, the 128 x 128 tile) simply
    come back without statistics, and asking the library directly is an error, not a silent no-op."""
    from v_express_amd import lib as L
    nb, hh, ww, c = 8, 32, 32, 320
    x = rnd(nb, hh, ww, c)
    wt = rnd(c, c, 3, 3, scale=(9 * c) ** -0.5, seed=1)
    bias = rnd(c, seed=2, dtype=torch.float32)
    w2d = wt.permute(0, 2, 3, 1).reshape(c, -1).contiguous()
    g = ops.ConvGeom(nb, hh, ww, 3, 3, 2, 1)
    out = ops.gemm(x.view(-1, c), w2d, bias, geom=g, gn=(32, g.h_out * g.w_out))
    check(out, _conv_ref(x, wt, bias, 2, 1, 0).reshape(-1, c), "downsample conv with GroupNorm partial sums")
    _gn_check(ops, out, nb, g.h_out * g.w_out, 32, "downsample conv")
    a, w = rnd(8 * 64, 2560), rnd(320, 2560, scale=2560 ** -0.5, seed=4)
    with ops.frame_rows(64):
        assert ops.gn_of(ops.gemm(a, w, bias, gn=(32, 64))) is None                 # split-K
    a, w = rnd(512, 128), rnd(960, 128, scale=0.1, seed=5)
    assert ops.gn_of(ops.gemm(a, w, None, gn=(32, 64))) is None                     # 30 channels per group
    a, w = rnd(512, 128), rnd(512, 128, scale=0.1, seed=6)
    assert ops.gn_of(ops.gemm(a, w, None, gn=(32, 64))) is None                     # the 128 x 128 tile
    assert ops.gn_of(ops.gemm(a, w, None, gn=(32, 64), out_f32=True)) is None
    p, _ = ops._base_params(a, w, None)
    o = torch.empty((512, 512), device="cuda", dtype=BF)
    ws = torch.empty(8 * 32 * 2, device="cuda")
    p.epi, p.out, p.ldc, p.alpha = L.VX_EPI_STORE, o.data_ptr(), 512, 1.0
    p.gn_ws, p.gn_groups, p.gn_hw = ws.data_ptr(), 32, 64
    assert ops._lib.vx_gemm_gn_slabs(p) == 0
    with pytest.raises(L.VxError):
        L.check(ops._lib.vx_gemm(p, ops._stream()), "vx_gemm")


# ------------------------------------------------------------------------------------------ cooperative two-way K split (ring_hint = 2)
def _coop_used(prof):
    return len(prof.records) > 0 and all(r[3].startswith("gemm_ring_kernel<") and r[3].endswith(",coop2>") for r in prof.records)


@pytest.mark.parametrize("frames,n,k,res", [(32, 1280, 5120, True), (32, 1280, 2560, False), (16, 1280, 5120, True),
                                            (64, 1280, 2560, True), (32, 1280, 2688, False), (15, 1280, 2560, True),
                                            (30, 1280, 5120, False)])
def test_gemm_ring_coop_split_linear(ops, monkeypatch, frames, n, k, res):
    """the 16x16-level launches (256 rows per frame, 128 tiles of 256 x 320 for a CFG pair) on the
    persistent kernel with the two K halves of a tile on two CUs that meet inside the launch (vx_gemm_params.ring_hint = 2;
    FF out-projection modules/attention FeedForward via mutual_self_attention.py:247, K = 5120).  Against fp32 math, against
    the 128 x 160 tiles (other summation order: bf16 ulps), half the batch / twice the batch = the same bits, and launch
    after launch on the same workspace (the flag words clean themselves)."""
    hw, m = 256, frames * 256
    monkeypatch.setattr(ops, "COOP_MIN_K", [2560])   # (the model's policy takes the split from K = 8192: the long 3x3 convs)
    a, w = rnd(m, k), rnd(n, k, scale=k ** -0.5, seed=1)
    bias = rnd(n, seed=2, dtype=torch.float32) + 0.3
    r = rnd(m, n, seed=3) if res else None
    # (15 / 30 frames: 120 / 240 work items = 15 / 30 per XCD - an odd count puts some partner pairs on DIFFERENT XCDs: the
    #  write-through + invalidate path of the rendezvous, which the model's own geometries never take)
    items = 2 if frames in (32, 30) else max(1, frames // 16)
    with ops.frame_rows(hw, items=items), ops.GemmProfile() as prof:
        out = ops.gemm(a, w, bias, residual=r, alpha=0.8)
    if (k // 64) % 2:                                # an odd number of 64-channel chunks cannot be halved: classic tiles
        assert not _coop_used(prof) and "gemm_ring" not in prof.records[0][3], prof.records[0][3]
        return
    assert _coop_used(prof), prof.records[0][3]
    ref = (a.float() @ w.float().t() + bias) * 0.8 + (r.float() if res else 0.0)
    check(out, ref, f"cooperative split {m}x{n}x{k}")
    with ops.frame_rows(hw, items=items):
        for _ in range(3):                           # same workspace, same bits
            assert torch.equal(ops.gemm(a, w, bias, residual=r, alpha=0.8), out)
        monkeypatch.setattr(ops, "RING_COOP", [False])
        with ops.GemmProfile() as prof2:
            classic = ops.gemm(a, w, bias, residual=r, alpha=0.8)
        monkeypatch.setattr(ops, "RING_COOP", [True])
    assert not _coop_used(prof2)
    d = (out.float() - classic.float()).abs()
    assert d.max().item() <= 2 ** -6 * ref.abs().max().item() and (d > 0).float().mean().item() < 0.2
    if items >= 2:                                   # one item alone: same kernel, same halves, same bits
        h = m // items
        with ops.frame_rows(hw, items=1), ops.GemmProfile() as prof3:
            one = ops.gemm(a[h:2 * h], w, bias, residual=None if r is None else r[h:2 * h], alpha=0.8)
        assert _coop_used(prof3) and torch.equal(one, out[h:2 * h])


@pytest.mark.parametrize("c1,c2,rowb,res", [(1280, 0, True, False), (1280, 1280, False, True), (640, 0, False, False)])
def test_gemm_ring_coop_split_conv(ops, monkeypatch, c1, c2, rowb, res):
    """The 16x16-level 3x3 convolutions of ResnetBlock3D (modules/resnet.py:223, 244; the up blocks' concatenated skip as a
    second source, modules/unet_3d_blocks.py:694) through the cooperative split: the halves are whole channel chunks, each
    walked taps-innermost; GroupNorm partial sums from the epilogue of whichever half arrives second."""
    nb, hh, ww, cout = 32, 16, 16, 1280
    cin = c1 + c2
    monkeypatch.setattr(ops, "COOP_MIN_K", [2560])
    x1 = torch.zeros(nb, hh + 2, ww + 2, c1, device="cuda", dtype=BF)
    x1[:, 1:-1, 1:-1] = rnd(nb, hh, ww, c1)
    x2 = None
    if c2:
        x2 = torch.zeros(nb, hh + 2, ww + 2, c2, device="cuda", dtype=BF)
        x2[:, 1:-1, 1:-1] = rnd(nb, hh, ww, c2, seed=7)
    wt = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=1)
    bias = rnd(cout, seed=2, dtype=torch.float32)
    w2d = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    g = ops.ConvGeom(nb, hh + 2, ww + 2, 3, 3, 1, 0)
    rows = hh * ww * 16
    rowbias = rnd(2, cout, seed=5, dtype=torch.float32) if rowb else None
    r = rnd(nb * hh * ww, cout, seed=6) if res else None
    with ops.frame_rows(hh * ww, items=2), ops.GemmProfile() as prof:
        out = ops.gemm(x1.view(-1, c1), w2d, bias, geom=g, a2=None if x2 is None else x2.view(-1, c2), rowbias=rowbias,
                       rows_per_group=rows if rowb else 0, residual=r, gn=(32, hh * ww))
        again = ops.gemm(x1.view(-1, c1), w2d, bias, geom=g, a2=None if x2 is None else x2.view(-1, c2), rowbias=rowbias,
                         rows_per_group=rows if rowb else 0, residual=r, gn=(32, hh * ww))
    assert _coop_used(prof), prof.records[0][3]
    x = x1 if x2 is None else torch.cat([x1, x2], dim=-1)
    ref = _conv_ref(x, wt, bias, 1, 0, 0).reshape(nb * hh * ww, cout)
    if rowb:
        ref = ref + rowbias.repeat_interleave(rows, 0)
    if res:
        ref = ref + r.float()
    check(out, ref, f"cooperative split conv {c1}+{c2}")
    assert torch.equal(out, again) and torch.equal(ops.gn_of(out).ws, ops.gn_of(again).ws)
    st = _gn_check(ops, out, nb, hh * ww, 32, "cooperative split conv")
    assert st.slabs == 2
    with ops.frame_rows(hh * ww, items=1):      # the conditional half alone
        one = ops.gemm(x1[16:].reshape(-1, c1), w2d, bias, geom=ops.ConvGeom(16, hh + 2, ww + 2, 3, 3, 1, 0),
                       a2=None if x2 is None else x2[16:].reshape(-1, c2),
                       rowbias=None if rowbias is None else rowbias[1:], rows_per_group=rows if rowb else 0,
                       residual=None if r is None else r[16 * hh * ww:], gn=(32, hh * ww))
    assert torch.equal(one, out[16 * hh * ww:]) and torch.equal(ops.gn_of(one).ws, st.ws[16:])


def test_gemm_ring_coop_split_refusals(ops):
    """The library answers for its own limits (vx_gemm_ring_coop_ok) and a forced request that cannot run is an error."""
    from v_express_amd import lib as L
    a, w = rnd(8192, 2560), rnd(1280, 2560, scale=0.02, seed=1)
    o = torch.empty((8192, 1280), device="cuda", dtype=BF)
    p, _ = ops._base_params(a, w, None)
    p.epi, p.out, p.ldc, p.alpha = L.VX_EPI_STORE, o.data_ptr(), 1280, 1.0
    assert ops._lib.vx_gemm_ring_coop_ok(p) == 1
    p2, _ = ops._base_params(a[:, :2496].contiguous(), w[:, :2496].contiguous(), None)      # 39 chunks
    p2.epi, p2.out, p2.ldc, p2.alpha = L.VX_EPI_STORE, o.data_ptr(), 1280, 1.0
    assert ops._lib.vx_gemm_ring_coop_ok(p2) == 0
    p2.ring_hint, p2.splitk, p2.coop_epoch = 2, 2, 1
    ws = torch.zeros(int(ops._lib.vx_gemm_splitk_ws_bytes(8192, 1280, 2)), device="cuda", dtype=torch.uint8)
    p2.splitk_ws = ws.data_ptr()
    with pytest.raises(L.VxError):
        L.check(ops._lib.vx_gemm(p2, ops._stream()), "vx_gemm")
    p.ring_hint, p.splitk, p.coop_epoch = 2, 2, 1      # no workspace
    with pytest.raises(L.VxError):
        L.check(ops._lib.vx_gemm(p, ops._stream()), "vx_gemm")
    p.splitk_ws, p.coop_epoch = ws.data_ptr(), 0       # a workspace, but no epoch (ABI 14: 1 <= coop_epoch < 2^27)
    with pytest.raises(L.VxError):
        L.check(ops._lib.vx_gemm(p, ops._stream()), "vx_gemm")


def test_gemm_ring_coop_split_survives_stale_rendezvous_words(ops):
    """with reset-to-zero flag words one stale word (a reader that gave up on its bounded poll, an aborted
    launch) made every later launch of the process add accumulators that were not written yet.  ABI 14: the words carry the
    launch's epoch and are never reset - a workspace whose words hold ANY older state gives the same bits as a fresh one."""
    from v_express_amd import lib as L
    m, n, k = 8192, 1280, 8192
    a, w = rnd(m, k), rnd(n, k, scale=k ** -0.5, seed=1)
    bias = rnd(n, seed=2, dtype=torch.float32)
    nbytes = int(ops._lib.vx_gemm_splitk_ws_bytes(m, n, 2))

    def run(ws, epoch):
        o = torch.empty((m, n), device="cuda", dtype=BF)
        p, _ = ops._base_params(a, w, None)
        p.epi, p.out, p.ldc, p.alpha, p.bias = L.VX_EPI_STORE, o.data_ptr(), n, 1.0, bias.data_ptr()
        p.ring_hint, p.splitk, p.splitk_ws, p.coop_epoch = 2, 2, ws.data_ptr(), epoch
        L.check(ops._lib.vx_gemm(p, ops._stream()), "vx_gemm")
        return o
    fresh = torch.zeros(nbytes, device="cuda", dtype=torch.uint8)
    ref = run(fresh, 1)
    check(ref, a.float() @ w.float().t() + bias, "cooperative split, fresh workspace")
    assert torch.equal(run(fresh, 2), ref) and torch.equal(run(fresh, 3), ref)
    # every word of the flag area left at an older epoch: "claimed" (word 0) and "ready" (word 1) of launches 5 / 6, and the
    # exchange area full of garbage - what aborted launches could leave at worst
    dirty = torch.zeros(nbytes, device="cuda", dtype=torch.uint8)
    flags = dirty[m * n * 4:].view(torch.int32)[:(m // 256) * (n // 320) * 8 * 2]    # [tile][wave][claimed, ready]
    dirty[:m * n * 4].view(torch.float32).fill_(float("nan"))
    flags[0::2] = 5
    flags[1::2] = (6 << 4) | 3
    assert torch.equal(run(dirty, 7), ref) and torch.equal(run(dirty, 8), ref)


@pytest.mark.parametrize("m", [128 * 3, 128 * 600])
def test_ff_fused_prototype_matches_the_two_launches(ops, m):
    """vx_ff_fused (round-4 prototype, C = 320): GEGLU feed-forward with the folded LayerNorm and the residual in ONE launch
    (diffusers FeedForward via modules/mutual_self_attention.py:247) against the product's two vx_gemm launches - the same
    arithmetic chain in the same order, so all but a few elements agree bit for bit - and against float32 math."""
    c, hidden = 320, 1280
    x = rnd(m, c) + 0.3
    w1 = rnd(2 * hidden, c, scale=c ** -0.5, seed=1)             # (already value / gate interleaved in blocks of 8)
    w2 = rnd(c, hidden, scale=hidden ** -0.5, seed=2)
    b1, b2 = rnd(2 * hidden, seed=3, dtype=torch.float32) * 0.3, rnd(c, seed=4, dtype=torch.float32) * 0.3
    colsum = w1.float().sum(dim=1).contiguous()
    stats = ops.row_stats(x)
    g = ops.geglu(x, w1, b1, ln=(stats, colsum))
    want = ops.gemm(g, w2, b2, residual=x)
    got = ops.ff_fused(x.clone(), w1, b1, colsum, stats, w2, b2)
    diff = (got.float() - want.float()).abs()
    frac = (got != want).float().mean().item()
    print(f"[ff_fused m={m}] max|diff| vs two launches {diff.max().item():.4g}, differing elements {100 * frac:.3g} %")
    assert diff.max().item() <= 2 ** -7 * want.float().abs().max().item() and frac < 0.02, (diff.max().item(), frac)
    xf = x.float()
    mean, var = xf.mean(dim=1, keepdim=True), xf.var(dim=1, unbiased=False, keepdim=True)
    p = ((xf - mean) * torch.rsqrt(var + 1e-5)) @ w1.float().t() + b1
    p = p.view(m, 2 * hidden // 16, 2, 8)
    h = (p[:, :, 0] * F.gelu(p[:, :, 1])).reshape(m, hidden)
    ref = xf + h.to(BF).float() @ w2.float().t() + b2
    check(got, ref, f"ff_fused {m} rows vs float32", rel=8e-3)


# ------------------------------------------------------------- round 4: the temporal attention block in one launch
def _tblock_problem(b, hw, seed=0, f=16):
    c, heads = 320, 8
    x = rnd(b * f * hw, c, seed=seed) * 1.5 + 0.3
    wqkv = rnd(3 * c, c, scale=c ** -0.5, seed=seed + 1)
    wo = rnd(c, c, scale=c ** -0.5, seed=seed + 2)
    bq = rnd(3 * c, seed=seed + 3, dtype=torch.float32) * 0.2
    bo = rnd(c, seed=seed + 4, dtype=torch.float32) * 0.2
    pe = rnd(24, 3 * c, seed=seed + 5, dtype=torch.float32) * 0.5
    colsum = wqkv.float().sum(dim=1).contiguous()
    return x, wqkv, wo, bq, bo, pe, colsum


@pytest.mark.parametrize("f", [16, 24])
def test_tblock_pack_matches_the_emulated_layout(ops, f):
    """vx_tblock_pack against tools/tblock_emulate.pack - the numpy statement of the fragment-major layouts that the
    lane-level emulation of the kernel (tests/test_host_logic.py) checks against plain float64 math."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import tblock_emulate as E
    from v_express_amd import lib as L
    _, wqkv, wo, bq, _, pe, colsum = _tblock_problem(1, 8)
    dev = wqkv.device
    nbytes = int(L.lib.vx_tblock_packed_bytes(f))
    assert nbytes == 32 * (20480 + (2048 if f == 16 else 4096))
    wqkv_t = torch.empty(nbytes // 2, device=dev, dtype=BF)
    wo_t = torch.empty(204800 // 2, device=dev, dtype=BF)
    cs = torch.empty(1024, device=dev, dtype=torch.float32)
    L.check(L.lib.vx_tblock_pack(wqkv.data_ptr(), bq.data_ptr(), colsum.data_ptr(), pe.data_ptr(), pe.stride(0),
                                 wo.data_ptr(), wqkv_t.data_ptr(), wo_t.data_ptr(), cs.data_ptr(), 320, 8, f,
                                 torch.cuda.current_stream().cuda_stream), "vx_tblock_pack")
    e_w, e_tab, e_wo, e_cs = E.pack(wqkv.float().cpu().numpy(), bq.cpu().numpy(), colsum.cpu().numpy(),
                                    pe[:f].cpu().numpy(), wo.float().cpu().numpy(), f)
    chunks = wqkv_t.view(torch.uint8).cpu().view(32, nbytes // 32)   # a chunk: 20480 B of bf16 weights, then its fp32 tables
    assert torch.equal(chunks[:, :20480].contiguous().view(BF).float(), torch.from_numpy(e_w).reshape(32, -1))
    assert torch.equal(chunks[:, 20480:].contiguous().view(torch.float32), torch.from_numpy(e_tab).reshape(32, -1))
    assert torch.equal(wo_t.float().cpu(), torch.from_numpy(e_wo).reshape(-1))
    assert torch.equal(cs.cpu(), torch.from_numpy(e_cs))


@pytest.mark.parametrize("b,hw,given_stats,f", [(1, 8, True, 16), (2, 64, False, 16), (2, 4096, True, 16), (2, 4096, False, 16),
                                                (3, 1160, False, 16), (1, 4, True, 24), (2, 64, False, 24), (2, 4096, True, 24),
                                                (3, 1156, False, 24)])
def test_tblock_fused_matches_the_three_launches(ops, b, hw, given_stats, f):
    """vx_tblock_fused (C = 320, 8 heads, 16 frames - or 24, the reference's default window, inference.py:67: two 16-row
    blocks per pixel, the padded key frames masked): LayerNorm-folded QKV projection + positional rows, attention over the
    frame axis, out-projection and residual in ONE launch (VersatileAttention inside TemporalTransformerBlock,
    modules/motion_module.py:243-256, :351-388) against the three launches it replaces - same rounding points, so the two
    differ only by the summation order inside the 16 x 16 products: a few bf16 ulps on a few elements - and against
    float32 math.  given_stats = False: the kernel takes the LayerNorm statistics from the rows it holds.
    (3, 1160): a tile count that is no multiple of the grid, pixels per item no power of two."""
    c, heads = 320, 8
    d = c // heads
    x, wqkv, wo, bq, bo, pe, colsum = _tblock_problem(b, hw, f=f)
    m = b * f * hw
    assert ops.tblock_fused_applies(c, heads, f, hw)
    stats = ops.row_stats(x)
    with ops.frame_rows(hw, items=b):
        qkv = ops.gemm(x, wqkv, bq, rowbias=pe[:f].repeat(b, 1).contiguous(), rows_per_group=hw, ln=(stats, colsum))
        a = ops.temporal_attention(qkv, b=b, f=f, hw=hw, heads=heads, head_dim=d)
        want = ops.gemm(a, wo, bo, residual=x)
    so = stats.clone() if given_stats else torch.full((m, 2), float("nan"), device="cuda")    # given: in place, as the model does
    got = ops.tblock_fused(x.clone(), wqkv, bq, colsum, pe, wo, bo, b=b, f=f, hw=hw, heads=heads,
                           stats=so if given_stats else None, stats_out=so)
    torch.cuda.synchronize()
    # stats_out: (mean, rstd) of the rows as stored, for the next LayerNorm fold
    g64 = got.double()
    mean, rstd = g64.mean(dim=1), torch.rsqrt(g64.var(dim=1, unbiased=False) + 1e-5)
    assert torch.isfinite(so).all()
    assert torch.allclose(so[:, 0].double(), mean, rtol=2e-5, atol=2e-6), (so[:, 0].double() - mean).abs().max()
    assert torch.allclose(so[:, 1].double(), rstd, rtol=1e-4), ((so[:, 1].double() - rstd) / rstd).abs().max()
    diff = (got.float() - want.float()).abs()
    frac = (got != want).float().mean().item()
    scale = want.float().abs().max().item()
    print(f"[tblock_fused f={f} b={b} hw={hw} stats={'given' if given_stats else 'own'}] max|diff| vs three launches "
          f"{diff.max().item():.4g} (max |out| {scale:.3g}), differing elements {100 * frac:.3g} %, "
          f"rel-L2 {(diff.norm() / want.float().norm()).item():.3g}")
    assert diff.max().item() <= 2 ** -5 * scale and (diff.norm() / want.float().norm()).item() <= 2e-3
    # float32 statement of the block
    xf = x.float()
    mean, var = xf.mean(dim=1, keepdim=True), xf.var(dim=1, unbiased=False, keepdim=True)
    ln = (xf - mean) * torch.rsqrt(var + 1e-5)
    q3 = ln @ wqkv.float().t() + bq + pe[:f].repeat(b, 1).repeat_interleave(hw, dim=0)
    q, k, v = (t.reshape(b, f, hw, heads, d).permute(0, 2, 3, 1, 4) for t in q3.to(BF).float().chunk(3, dim=-1))
    o = F.scaled_dot_product_attention(q, k, v).permute(0, 3, 1, 2, 4).reshape(m, c)
    ref = xf + o.to(BF).float() @ wo.float().t() + bo
    check(got, ref, f"tblock_fused f={f} b={b} hw={hw} vs float32", rel=8e-3)


# ------------------------------------------------------------- round 4: row statistics in two parts (n = k = 640)
def _half_sums(x):
    x64 = x.double()
    h = x64.shape[1] // 2
    return torch.stack([x64[:, :h].sum(1), (x64[:, :h] ** 2).sum(1), x64[:, h:].sum(1), (x64[:, h:] ** 2).sum(1)], dim=1)


@pytest.mark.parametrize("m,k,res,ring", [(256 * 384, 640, True, True), (256 * 100, 2560, True, True), (256 * 128, 640, False, True),
                                          (3000, 640, True, False)])
def test_gemm_row_stats_two_parts(ops, m, k, res, ring):
    """vx_gemm_params.row_stats_parts = 2 (n = 640): (sum, sum of squares) of each half of every STORED bf16 row, from the
    epilogue of the persistent kernel's two column tiles (no pass re-reads the tensor) or - a launch off that kernel - from
    vx_row_stats_parts; against float64 sums of the tensor the launch wrote.  The output must not change; a launch over
    half the rows gives the same bits."""
    n = 640
    a, w = rnd(m, k), rnd(n, k, scale=k ** -0.5, seed=1)
    bias = rnd(n, seed=2, dtype=torch.float32) + 0.5
    r = rnd(m, n, seed=3) if res else None
    st = torch.full((m, 4), float("nan"), device="cuda")
    with ops.GemmProfile() as prof:
        out = ops.gemm(a, w, bias, residual=r, stats_out=st)
    assert _ring_used(prof) == ring, prof.records[0][3]
    assert torch.equal(out, ops.gemm(a, w, bias, residual=r))
    want = _half_sums(out)
    assert torch.isfinite(st).all()
    assert torch.allclose(st.double(), want, rtol=3e-6, atol=2e-3), (st.double() - want).abs().max()
    if ring and (m // 2) % 256 == 0 and (m // 2 // 256) * 2 >= 192:          # (the half launch stays on the persistent kernel)
        st2 = torch.empty((m // 2, 4), device="cuda")
        ops.gemm(a[:m // 2], w, bias, residual=None if r is None else r[:m // 2], stats_out=st2)
        assert torch.equal(st2, st[:m // 2])
    st3 = ops.row_stats(out, out=torch.empty((m, 4), device="cuda"))          # vx_row_stats_parts itself
    assert torch.allclose(st3.double(), want, rtol=3e-6, atol=2e-3)


@pytest.mark.parametrize("m,n,kind", [(256 * 128, 640, "store"), (256 * 128, 1920, "store"), (256 * 64, 5120, "geglu"),
                                      (256 * 128, 1920, "split"), (2000, 640, "store")])
def test_layernorm_fold_from_two_part_statistics(ops, m, n, kind):
    """vx_gemm_params.ln_stats_parts = 2 (k = 640): the folded-LayerNorm epilogue of the persistent kernel finishes the
    two-part sums itself (STORE and GEGLU); launches off that kernel (the QKV split of the classic tile, small problems)
    get them finished by vx_row_stats_finalize first.  Against the same launch fed (mean, rstd) of vx_row_stats: the two
    differ only by the one-pass variance (rstd within 1e-4)."""
    k = 640
    x = rnd(m, k) * 1.3 + 0.4
    w = rnd(n, k, scale=k ** -0.5, seed=1)
    b = rnd(n, seed=2, dtype=torch.float32) * 0.2
    colsum = w.float().sum(dim=1).contiguous()
    st2 = ops.row_stats(x)
    st4 = ops.row_stats(x, out=torch.empty((m, 4), device="cuda"))
    if kind == "store":
        got, want = ops.gemm(x, w, b, ln=(st4, colsum)), ops.gemm(x, w, b, ln=(st2, colsum))
    elif kind == "geglu":
        got, want = ops.geglu(x, w, b, ln=(st4, colsum)), ops.geglu(x, w, b, ln=(st2, colsum))
    else:
        heads, seq = 8, 64
        d = (n // 3) // heads

        def run(st):
            q = torch.empty((m, n // 3), device="cuda", dtype=BF)
            kk = torch.empty((m, n // 3), device="cuda", dtype=BF)
            vt = ops.alloc_vt(m // seq, heads, d, seq, "cuda")
            ops.gemm_split(x, w, b, [("rows", q), ("rows", kk), ("vt", vt)], part_cols=n // 3, seq_len=seq, head_dim=d,
                           ln=(st, colsum))
            return torch.cat([q.float(), kk.float(), vt[..., :seq].float().permute(0, 3, 1, 2).reshape(m, n // 3)], dim=1)
        got, want = run(st4), run(st2)
    diff = (got.float() - want.float()).abs()
    frac = (got != want).float().mean().item()
    print(f"[ln fold, two-part statistics {kind} m={m} n={n}] max|diff| {diff.max().item():.4g}, differing {100 * frac:.3g} %")
    assert diff.max().item() <= 2 ** -6 * want.float().abs().max().item() and frac < 0.02
