"""The reference's default arithmetic type (inference.py:44,150-151: `--dtype fp16`) on the MI355X: a float16 model runs
libvexpress_hip_f16.so - the SAME kernel sources compiled with -DVX_ELEM_F16 (IEEE half storage, v_mfma_f32_*_f16, fp32
accumulation), the same C ABI - selected by the model's dtype (module_base.DeviceModule, lib.element_type).

Tolerances: 11 instead of 8 mantissa bits, so every bf16 bound of tests/test_gpu_kernels.py is divided by 8 here:
kernels max|err| <= 2^-10 max|ref|, relative L2 <= 7.5e-4 (attention / GroupNorm 2^-9 / 1.25e-3); a CFG UNet3D forward at
SD-1.5 widths against the reference's own fp32 output relative L2 <= 4e-3 (bf16: 3e-2 allowed, 1.15e-2 measured).
The whole kernel test file runs in this mode with VX_TEST_ELEM=f16 (tools/gpu_job.sh tests16); the cases below always run.
"""
import os

import pytest
import torch
import torch.nn.functional as F

import cases

pytestmark = pytest.mark.gpu
H = torch.float16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from v_express_amd import lib as L, ops as o
    assert L.lib_f16().vx_element_type() == b"f16"
    with L.element_type(H):
        yield o


def rnd(*shape, scale=1.0, seed=0, dtype=H):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to("cuda").to(dtype)


def check(got, ref, what, rel=7.5e-4, mx=2 ** -10):
    assert got.dtype == H, f"{what}: output dtype {got.dtype}"
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape and torch.isfinite(got).all(), what
    err = (got - ref).abs()
    scale = ref.abs().max().item() + 1e-12
    rl2 = (err.pow(2).sum().sqrt() / (ref.pow(2).sum().sqrt() + 1e-12)).item()
    print(f"[f16 {what}] max|err|/max|ref| = {err.max().item() / scale:.3g}  relL2 = {rl2:.3g}")
    assert err.max().item() <= mx * scale + 1e-6 and rl2 <= rel, (what, err.max().item() / scale, rl2)


def test_f16_library_is_the_one_that_runs(ops):
    from v_express_amd import lib as L
    assert L.current() is L.lib_f16() and ops.BF16 is H
    a, w = rnd(512, 320), rnd(320, 320, scale=320 ** -0.5, seed=1)
    out = ops.gemm(a, w)
    assert out.dtype == H
    with pytest.raises(TypeError):
        ops.gemm(a.to(torch.bfloat16), w)                       # a bfloat16 tensor has no business in this library
    with L.element_type(torch.bfloat16):                        # ... and the bf16 library is one `with` away
        assert ops.gemm(a.to(torch.bfloat16), w.to(torch.bfloat16)).dtype == torch.bfloat16


@pytest.mark.parametrize("m,n,k,items", [(256 * 200, 640, 640, 2), (131, 1280, 640, None), (8192, 1280, 1280, 2),
                                         (256 * 64, 320, 2880, 2)])
def test_f16_gemm_persistent_and_classic_tiles(ops, m, n, k, items):
    """vx_gemm (F.linear / conv as GEMM, modules/resnet.py:9-17 & diffusers Attention / FeedForward linears): the persistent
    ring kernel and the classic tiles, bias + SiLU-free STORE epilogue with residual."""
    a, w = rnd(m, k), rnd(n, k, scale=k ** -0.5, seed=1)
    bias, res = rnd(n, seed=2, dtype=torch.float32), rnd(m, n, seed=3)
    ref = res.float() + 0.5 * (a.float() @ w.float().t() + bias)
    if items:
        with ops.frame_rows(m // (16 * items), items=items):
            out = ops.gemm(a, w, bias, residual=res, alpha=0.5)
    else:
        out = ops.gemm(a, w, bias, residual=res, alpha=0.5)
    check(out, ref, f"gemm {m}x{n}x{k}")


def test_f16_conv3x3_and_geglu(ops):
    """InflatedConv3d 3x3 (modules/resnet.py:9-17) on the zero-bordered image and the GEGLU projection
    (diffusers FeedForward(activation_fn='geglu'), modules/mutual_self_attention.py:247)."""
    nb, hh, ww, cin, cout = 4, 32, 32, 320, 640
    x = rnd(nb, hh, ww, cin)
    wt = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=1)
    bias = rnd(cout, seed=2, dtype=torch.float32)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), bias, padding=1).permute(0, 2, 3, 1)
    xp = F.pad(x, (0, 0, 1, 1, 1, 1)).contiguous()
    w2d = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    out = ops.gemm(xp.view(-1, cin), w2d, bias, geom=ops.ConvGeom(nb, hh + 2, ww + 2, 3, 3, 1, 0))
    check(out.view(nb, hh, ww, cout), ref, "conv3x3 pad-0 on a bordered image")
    m, c, hidden = 256 * 16, 640, 2560
    a = rnd(m, c)
    from v_express_amd.weights import geglu_interleave
    w1, b1 = rnd(2 * hidden, c, scale=c ** -0.5, seed=3), rnd(2 * hidden, seed=4, dtype=torch.float32)
    wi, bi = geglu_interleave(w1), geglu_interleave(b1)
    p = a.float() @ w1.float().t() + b1
    ref = p[:, :hidden] * F.gelu(p[:, hidden:])
    check(ops.geglu(a, wi, bi), ref, "geglu projection")


def test_f16_groupnorm_layernorm(ops):
    frames, hw, c, groups = 4, 1024, 640, 32
    x = rnd(frames, hw, c) * 2 + 0.5
    g, b = rnd(c, seed=1, dtype=torch.float32) * 0.1 + 1, rnd(c, seed=2, dtype=torch.float32) * 0.1
    out = ops.groupnorm(x, g, b, frames=frames, hw=hw, groups=groups, eps=1e-5, silu=True)
    ref = F.silu(F.group_norm(x.float().permute(0, 2, 1), groups, g, b, 1e-5)).permute(0, 2, 1)
    check(out, ref, "groupnorm + silu", rel=1.25e-3, mx=2 ** -9)
    y = ops.layernorm(x.view(-1, c), g, b)
    check(y, F.layer_norm(x.float().view(-1, c), (c,), g, b, 1e-5), "layernorm", rel=1.25e-3, mx=2 ** -9)


@pytest.mark.parametrize("batch,heads,n,d,qscale", [(2, 8, 4096, 40, 1.0), (2, 8, 1024, 80, 1.0), (1, 8, 4096, 40, 6.0),
                                                    (2, 8, 100, 160, 1.0)])
def test_f16_attention(ops, batch, heads, n, d, qscale):
    """F.scaled_dot_product_attention of AttnProcessor2_0 (modules/mutual_self_attention.py:177-224).  d = 40 runs the bounded
    softmax (attn3): in IEEE half its shift is lowered by 14 log2 units and rows that sit far under their Cauchy-Schwarz
    bound go to the exact recompute - qscale = 6 makes the logits large enough (|s| ~ 35 log2 units) to take that path."""
    c = heads * d
    q, k, v = rnd(batch * n, c) * qscale, rnd(batch * n, c, seed=1), rnd(batch * n, c, seed=2)
    vt = ops.alloc_vt(batch, heads, d, n, "cuda")
    vt[..., :n] = v.view(batch, n, heads, d).permute(0, 2, 3, 1)
    out = ops.attention(q, k, vt, batch=batch, heads=heads, n_q=n, n_kv=n, head_dim=d)     # (d = 40: kmax computed inside)
    sh = lambda t: t.view(batch, n, heads, d).transpose(1, 2).float()          # noqa: E731
    ref = F.scaled_dot_product_attention(sh(q), sh(k), sh(v)).transpose(1, 2).reshape(batch * n, c)
    check(out, ref, f"attention n={n} d={d} qscale={qscale}", rel=1.25e-3, mx=2 ** -9)


def test_f16_one_launch_blocks(ops):
    """vx_ff_fused / vx_tblock_fused (the 64x64 level's GEGLU feed-forward and temporal attention block in one launch each:
    modules/mutual_self_attention.py:247, modules/motion_module.py:243-256,351-388) against float32 math."""
    g = torch.Generator().manual_seed(11)

    def r(*shape, scale=1.0, dtype=H):
        return (torch.randn(*shape, generator=g) * scale).to("cuda").to(dtype)
    c, heads, f, hw, b, hidden = 320, 8, 16, 64, 1, 1280
    d, m = c // heads, b * f * hw
    x = r(m, c) * 1.5 + 0.3
    xf = x.float()
    ln = (xf - xf.mean(1, keepdim=True)) * torch.rsqrt(xf.var(1, unbiased=False, keepdim=True) + 1e-5)
    w1, w2 = r(2 * hidden, c, scale=c ** -0.5), r(c, hidden, scale=hidden ** -0.5)
    b1, b2 = r(2 * hidden, dtype=torch.float32) * 0.3, r(c, dtype=torch.float32) * 0.3
    got = ops.ff_fused(x.clone(), w1, b1, w1.float().sum(1).contiguous(), ops.row_stats(x), w2, b2)
    p = (ln @ w1.float().t() + b1).view(m, 2 * hidden // 16, 2, 8)
    ref = xf + (p[:, :, 0] * F.gelu(p[:, :, 1])).reshape(m, hidden).to(H).float() @ w2.float().t() + b2
    check(got, ref, "ff_fused", rel=1.25e-3, mx=2 ** -9)
    wqkv, wo = r(3 * c, c, scale=c ** -0.5), r(c, c, scale=c ** -0.5)
    bq, bo = r(3 * c, dtype=torch.float32) * 0.2, r(c, dtype=torch.float32) * 0.2
    pe = r(f, 3 * c, dtype=torch.float32) * 0.5
    got = ops.tblock_fused(x.clone(), wqkv, bq, wqkv.float().sum(1).contiguous(), pe, wo, bo, b=b, f=f, hw=hw, heads=heads)
    q3 = (ln @ wqkv.float().t() + bq + pe.repeat(b, 1).repeat_interleave(hw, dim=0)).to(H).float()
    q, k, v = (t.reshape(b, f, hw, heads, d).permute(0, 2, 3, 1, 4) for t in q3.chunk(3, dim=-1))
    o = F.scaled_dot_product_attention(q, k, v).permute(0, 3, 1, 2, 4).reshape(m, c)
    ref = xf + o.to(H).float() @ wo.float().t() + bo
    check(got, ref, "tblock_fused", rel=1.25e-3, mx=2 ** -9)


def _rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).sum().sqrt() / (b.pow(2).sum().sqrt() + 1e-12)).item()


def test_f16_small_unet_forward_and_vae_vs_oracle():
    """A float16 UNet3D + ReferenceNet + VAE decoder (every block type live, small widths) against the fp32 oracle: the model
    classes pick the IEEE-half library from their dtype; relative L2 8x under the bf16 bound of __graft_entry__.smoke()."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import oracle
    import v_express_amd as vx
    from oracle import unet as OU
    from oracle import vae as OV
    from v_express_amd import lib as L, synth
    kw = dict(block_out_channels=(64, 128, 256, 256))
    cfg, ocfg = synth.UNetConfig(**kw), oracle.UNetConfig(**kw)
    sd3, sd2 = synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg)
    Fr, h, w, t = 4, 8, 8, 519
    inp = synth.synthetic_inputs(cfg, Fr, h, w)
    unet = vx.UNet3DConditionModel(cfg).to("cuda").half()
    refnet = vx.UNet2DConditionModel(cfg).to("cuda").half()
    unet.load_state_dict(sd3)
    refnet.load_state_dict(sd2)
    writer = vx.ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    reader = vx.ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                          reference_attention_weight=0.95, audio_attention_weight=3.0)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
    reader.update(writer, True)
    assert all(kv is None or kv[0].dtype == H for rows in unet.banks.values() for kv in rows)
    x = inp["latents"].repeat(2, 1, 1, 1, 1)
    ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
    before = L.ELEM[0]
    with L.element_type(torch.bfloat16):                   # whatever is in force outside, the model runs under ITS element type
        got = unet(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0].float().cpu()
        assert L.ELEM[0] is torch.bfloat16                 # ... and leaves the caller's in force
    assert L.ELEM[0] is before
    banks = OU.reader_banks(OU.refnet_banks(sd2, ocfg, inp["ref_latents"]))
    ref = OU.unet3d_forward(sd3, ocfg, x, t, ehs, inp["kps_features"], banks, 0.95, 3.0)
    rel = _rel_l2(got, ref)
    vkw = dict(block_out_channels=(32, 64, 128, 128))
    vcfg = synth.VaeConfig(**vkw)
    sdv = synth.vae_decoder_state_dict(vcfg)
    vae = vx.AutoencoderKLDecoder(vcfg).to("cuda").half()
    vae.load_state_dict(sdv)
    z = inp["latents"][0, :, :2].permute(1, 0, 2, 3).contiguous()
    vrel = _rel_l2(vae.decode(z).sample, OV.vae_decode(sdv, oracle.VaeConfig(**vkw), z))
    print(f"[f16 small models] UNet3D relL2 = {rel:.3g} (bf16 smoke: ~1e-2), VAE relL2 = {vrel:.3g} (bf16: ~2.7e-2)")
    assert torch.isfinite(got).all() and rel <= 4e-3 and vrel <= 4e-3, (rel, vrel)


def test_f16_fullsize_forward_and_25_step_call_vs_reference_golden():
    """VERDICT r05 item 7 at BASELINE configs[1]'s size, the float16 models against the REFERENCE's own fp32 run
    (tests/golden/fullsize_F16_512.pt): (1) one CFG forward (SD-1.5 widths, 16 frames at 64x64 latents, t = 999) at relative
    L2 <= 4e-3 - the bound the bf16 build cannot meet (1.15e-2); (2) `VExpressPipeline.__call__`: 25 DDIM steps + decode,
    latents after steps 1 / 5 / 13 / 25 within the bf16 bounds of tests/test_gpu_fullsize.py DIVIDED BY 4
    (7.5e-4 / 2.5e-3 / 5e-3 / 7.5e-3), decoded frames PSNR >= 50 dB; and no activation of the path leaves half's range
    (every output finite).  pipelines/v_express_pipeline.py:526-589,152-166."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    gold_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_F16_512.pt")
    import ref_import as R
    import v_express_amd as vx
    from v_express_amd import synth
    cfg, vcfg = cases.unet_cfg(cases.FULL), synth.VaeConfig()
    unet = vx.UNet3DConditionModel(cfg).to("cuda").half()
    refnet = vx.UNet2DConditionModel(cfg).to("cuda").half()
    unet.load_state_dict(synth.unet3d_state_dict(cfg), strict=True)
    unet.release_raw_weights()
    refnet.load_state_dict(synth.refnet_state_dict(cfg), strict=True)
    refnet.release_raw_weights()
    vae = vx.AutoencoderKLDecoder(vcfg).to("cuda").half()
    vae.load_state_dict(synth.vae_decoder_state_dict(vcfg))
    Fr, cf, co, steps = cases.FULLSIZE_CASE
    inp = synth.synthetic_inputs(cfg, Fr, 64, 64)
    gold = torch.load(gold_path, weights_only=False)
    writer = vx.ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    reader = vx.ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                          reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
    reader.update(writer, True)
    x = inp["latents"].repeat(2, 1, 1, 1, 1)
    ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
    got = unet(x, 999, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
    reader.clear()
    writer.clear()
    want = gold["pred_step0"]
    r = _rel_l2(got, want)
    print(f"[f16 fullsize f=16 forward, t=999] relL2 = {r:.4g} (bf16 build: 1.15e-2)")
    assert got.shape == want.shape and torch.isfinite(got).all() and r <= 4e-3, r
    pipe = vx.VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet,
                               scheduler=vx.DDIMScheduler(**R.NOISE_SCHEDULER_KWARGS))
    assert pipe.dtype == H
    trace = {}
    video = pipe(None, None, None, 512, 512, Fr, steps, cases.GUIDANCE, context_frames=cf, context_overlap=co,
                 reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD,
                 reference_latents=inp["ref_latents"], kps_features=inp["kps_features"],
                 audio_embeddings=inp["audio_embeddings"], latents=inp["latents"],
                 callback=lambda i, t, l: trace.__setitem__(i, l.detach().cpu().clone()) if i in (0, 4, 12, 24) else None)
    assert video.shape == (1, 3, Fr, 512, 512) and video.dtype == torch.float32 and torch.isfinite(video).all()
    bounds = {0: ("latents_step0", 7.5e-4), 4: ("latents_step4", 2.5e-3), 12: ("latents_step12", 5e-3), 24: ("latents", 7.5e-3)}
    bad = []
    for i, (key, bound) in bounds.items():
        rr = _rel_l2(trace[i], gold[key])
        print(f"[f16 fullsize loop] after step {i + 1:2d}: relL2 = {rr:.4g}  (bound {bound:.1e}; bf16 bound {4 * bound:.1e})")
        if not rr <= bound:
            bad.append((i + 1, rr, bound))
    frames = list(gold["video_frames"])
    mse = (video[:, :, frames].float() - gold["video_f16"].float()).pow(2).mean().item()
    p = 10 * torch.log10(torch.tensor(1.0 / max(mse, 1e-12))).item()
    print(f"[f16 fullsize loop] decoded frames {frames}: PSNR = {p:.1f} dB (bf16 build: 50.8)")
    assert not bad and p >= 50.0, (bad, p)
