"""Model-level parity on a real MI355X: the HIP UNet3D / ReferenceNet / VAE / pipeline against the fp32 CPU oracle
(oracle/, itself pinned to the reference by tests/golden/) on identical seeded weights and inputs.

Stated tolerances (bf16 storage + fp32 accumulation through ~200 sequential layers vs an fp32 oracle):
  banks (ReferenceNet, ~60 layers)      relative L2 <= 3e-2 (deepest bank measures ~2e-2: bf16 storage noise)
  one UNet3D CFG forward                relative L2 <= 3e-2, cosine >= 0.999
  N-step loop latents                   relative L2 <= 5e-2, cosine >= 0.998
  decoded frames                        mean abs error <= 2e-2 (range [0,1]), PSNR >= 30 dB
"""
import os

import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).sum().sqrt() / (b.pow(2).sum().sqrt() + 1e-12)).item()


def cosine(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-12)).item()


def build_models(kw, sd3, sd2):
    from v_express_amd import ReferenceAttentionControl, UNet2DConditionModel, UNet3DConditionModel
    cfg = cases.unet_cfg(kw)
    unet = UNet3DConditionModel(cfg).to("cuda")
    refnet = UNet2DConditionModel(cfg).to("cuda")
    unet.load_state_dict(sd3, strict=True)
    refnet.load_state_dict(sd2, strict=True)
    return unet, refnet


@pytest.mark.parametrize("name", ["small_f4_8x8", "small_f8_16x8", "full_f4_8x8"])
def test_unet_forward_vs_oracle_and_golden(name):
    _need_gpu()
    from oracle import unet as OU
    from v_express_amd import ReferenceAttentionControl, synth
    kw, F, h, w, t = cases.FORWARD_CASES[name]
    cfg, ocfg = cases.unet_cfg(kw), cases.oracle_cfg(kw)
    sd3, sd2 = synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg)
    inp = synth.synthetic_inputs(cfg, F, h, w)
    unet, refnet = build_models(kw, sd3, sd2)
    writer = ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    reader = ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                       reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
    obanks = OU.refnet_banks(sd2, ocfg, inp["ref_latents"])
    assert sorted(refnet.banks) == sorted(obanks)
    worst = max((rel_l2(refnet.banks[k].view_as(obanks[k][0]), obanks[k][0]), k) for k in obanks)
    print(f"[{name}] worst bank relL2={worst[0]:.4g} ({worst[1]})")
    assert worst[0] <= 3e-2, f"bank parity: worst relL2 {worst}"
    reader.update(writer, True)
    x = inp["latents"].repeat(2, 1, 1, 1, 1)
    ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
    got = unet(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
    ref = OU.unet3d_forward(sd3, ocfg, x, t, ehs, inp["kps_features"], OU.reader_banks(obanks), cases.W_REF,
                            cases.W_AUD)
    gold = torch.load(os.path.join(GOLD, f"forward_{name}.pt"), weights_only=False)["pred"]
    assert (ref - gold).abs().max().item() < 2e-5, "oracle drifted from the reference golden"
    r, c = rel_l2(got, gold), cosine(got, gold)
    print(f"[{name}] relL2={r:.4g} cosine={c:.6f}")
    assert torch.isfinite(got).all()
    assert r <= 3e-2 and c >= 0.999, f"UNet3D forward parity: relL2={r:.4g} cosine={c:.6f}"
    # a lone CFG half (as another GPU would run it) must equal the matching half of the batched call
    from v_express_amd import ops
    for half in (0, 1):
        xin = ops.ncfhw_to_nhwc(x[half:half + 1].cuda(), 8)
        e = ehs[half * F:(half + 1) * F].reshape(-1, 768).to("cuda", torch.bfloat16).contiguous()
        kp = ops.ncfhw_to_nhwc(inp["kps_features"][half:half + 1].cuda(), cfg.block_out_channels[0])
        o = unet.forward_tokens(xin, t, e, kp, b=1, f=F, H=h, W=w, batch_rows=[half])
        y = ops.nhwc_to_ncfhw(o, 1, 4, F, h, w)
        assert torch.equal(y[0].cpu(), got[half].float().cpu()), f"CFG half {half} is not bit-identical when run alone"


@pytest.mark.parametrize("name", ["small_f4_8x8", "full_f4_8x8"])
def test_unet_forward_fp8_projections_vs_golden(name):
    """BASELINE.json configs[4] ("fp8 MFMA QKV/out-proj"): `unet.fp8_projections = True` sends the q / k / v / out
    projections of attn1, attn1_5, attn2 and both temporal attentions through per-row e4m3 operands on
    v_mfma_scale_f32_16x16x128_f8f6f4 (fp32 accumulation, bf16 results).  The reference has no fp8 mode; the stated
    tolerance of this mode against the fp32 reference golden is rel-L2 <= 6e-2, cosine >= 0.998 for one CFG forward
    (bf16 mode: 3e-2 / 0.999), and it must actually differ from the bf16 path."""
    _need_gpu()
    from v_express_amd import ReferenceAttentionControl, synth
    kw, F, h, w, t = cases.FORWARD_CASES[name]
    cfg = cases.unet_cfg(kw)
    unet, refnet = build_models(kw, synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg))
    inp = synth.synthetic_inputs(cfg, F, h, w)
    writer = ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    reader = ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                       reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
    reader.update(writer, True)
    x = inp["latents"].repeat(2, 1, 1, 1, 1)
    ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
    bf = unet(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
    unet.fp8_projections = True
    got = unet(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
    gold = torch.load(os.path.join(GOLD, f"forward_{name}.pt"), weights_only=False)["pred"]
    r, c = rel_l2(got, gold), cosine(got, gold)
    print(f"[{name}] fp8 projections: relL2={r:.4g} cosine={c:.6f}   (bf16: relL2={rel_l2(bf, gold):.4g})")
    assert torch.isfinite(got).all() and not torch.equal(got, bf)
    assert r <= 6e-2 and c >= 0.998, (r, c)


@pytest.mark.parametrize("latent", [64, 96])
def test_full_size_unet_forward_vs_oracle(latent):
    """latent = 96: the 768x768 geometry of BASELINE.json configs[4] (9216 / 2304 / 576 / 144 tokens per level: the
    pre-padded convs leave the ring kernel there, the attention kernels run ragged query blocks), same check.
    latent = 64: BASELINE.json configs[1] geometry against the oracle DIRECTLY: SD-1.5 widths, 512x512 (64x64 latents: 4096 tokens,
    head dims 40 / 80 / 160), CFG batch of 2 - the shapes on which the ring GEMM, the pre-padded convs, attn2 and the
    split-K path actually run - at f = 2 frames, the sample bench.py's cpu_baseline leg times (the fp32 oracle needs
    ~20 s per forward on the host cores; 16 frames would take minutes).  ReferenceNet banks and one UNet3D CFG forward,
    same tolerances as the small cases."""
    _need_gpu()
    from oracle import unet as OU
    from v_express_amd import ReferenceAttentionControl, synth
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(64, os.cpu_count() or 1))           # the oracle is slower on all 256 hardware threads
    try:
        kw, F, h, w, t = cases.FULL, 2, latent, latent, 519
        cfg, ocfg = cases.unet_cfg(kw), cases.oracle_cfg(kw)
        sd3, sd2 = synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg)
        inp = synth.synthetic_inputs(cfg, F, h, w)
        unet, refnet = build_models(kw, sd3, sd2)
        writer = ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
        reader = ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                           reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD)
        refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
        with torch.no_grad():
            obanks = OU.refnet_banks(sd2, ocfg, inp["ref_latents"])
        worst = max((rel_l2(refnet.banks[k].view_as(obanks[k][0]), obanks[k][0]), k) for k in obanks)
        assert worst[0] <= 3e-2, f"bank parity at {latent}x{latent} latents: worst relL2 {worst}"
        reader.update(writer, True)
        x = inp["latents"].repeat(2, 1, 1, 1, 1)
        ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
        got = unet(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
        with torch.no_grad():
            ref = OU.unet3d_forward(sd3, ocfg, x, t, ehs, inp["kps_features"], OU.reader_banks(obanks), cases.W_REF,
                                    cases.W_AUD)
        r, c = rel_l2(got, ref), cosine(got, ref)
        print(f"[full {8 * latent}x{8 * latent} f=2] worst bank relL2={worst[0]:.4g} ({worst[1]}); forward relL2={r:.4g} "
              f"cosine={c:.6f}")
        assert torch.isfinite(got).all() and r <= 3e-2 and c >= 0.999, (r, c)
        # BASELINE.json configs[4]'s fp8 q / k / v / out projections on the same inputs against the same oracle output
        # (latent = 96 is that configuration's own 768x768 geometry); stated tolerance of the mode: 6e-2 / 0.998
        unet.fp8_projections = True
        got8 = unet(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
        unet.fp8_projections = False
        r8, c8 = rel_l2(got8, ref), cosine(got8, ref)
        print(f"[full {8 * latent}x{8 * latent} f=2] fp8 projections: forward relL2={r8:.4g} cosine={c8:.6f}")
        assert torch.isfinite(got8).all() and not torch.equal(got8, got) and r8 <= 6e-2 and c8 >= 0.998, (r8, c8)
    finally:
        torch.set_num_threads(nthreads)


@pytest.mark.parametrize("name", list(cases.PIPELINE_CASES))
def test_pipeline_vs_reference_golden(name):
    _need_gpu()
    from v_express_amd import (AutoencoderKLDecoder, DDIMScheduler, UNet2DConditionModel, UNet3DConditionModel,
                               VExpressPipeline, synth)
    import ref_import as R
    Fn, cf, co, steps = cases.PIPELINE_CASES[name]
    kw = cases.SMALL
    cfg = cases.unet_cfg(kw)
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    unet, refnet = build_models(kw, synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg))
    vae = AutoencoderKLDecoder(vcfg).to("cuda")
    vae.load_state_dict(synth.vae_decoder_state_dict(vcfg))
    pipe = VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet,
                            scheduler=DDIMScheduler(**R.NOISE_SCHEDULER_KWARGS))
    inp = synth.synthetic_inputs(cfg, Fn, 8, 8)
    trace = []
    video = pipe(None, None, None, 64, 64, Fn, steps, cases.GUIDANCE, context_frames=cf, context_overlap=co,
                 reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD,
                 reference_latents=inp["ref_latents"], kps_features=inp["kps_features"],
                 audio_embeddings=inp["audio_embeddings"], latents=inp["latents"],
                 callback=lambda i, t, l: trace.append(l.detach().cpu().clone()))
    g = torch.load(os.path.join(GOLD, f"pipeline_{name}.pt"), weights_only=False)
    assert video.shape == g["video_f16"].shape and video.device.type == "cpu" and video.dtype == torch.float32
    r0, r1 = rel_l2(trace[0], g["latents_step0"]), rel_l2(trace[-1], g["latents"])
    c1 = cosine(trace[-1], g["latents"])
    mae = (video - g["video_f16"].float()).abs().mean().item()
    mse = (video - g["video_f16"].float()).pow(2).mean().item()
    psnr = 10 * torch.log10(torch.tensor(1.0 / max(mse, 1e-12))).item()
    print(f"[{name}] step0 relL2={r0:.4g} final relL2={r1:.4g} cos={c1:.6f} video MAE={mae:.4g} PSNR={psnr:.1f} dB")
    assert r0 <= 3e-2 and r1 <= 5e-2 and c1 >= 0.998, (r0, r1, c1)
    assert mae <= 2e-2 and psnr >= 30.0, (mae, psnr)
    assert video.min().item() >= 0.0 and video.max().item() <= 1.0


def test_merged_unet_calls_are_bit_identical_on_the_gpu():
    """`VExpressPipeline.units_per_call`: 2 (one window per UNet call) vs the default 4 vs 6 (consecutive windows merged
    into one batch) on the real kernels: batch rows never influence each other, so the clips must be bit-identical."""
    _need_gpu()
    from v_express_amd import AutoencoderKLDecoder, DDIMScheduler, VExpressPipeline, synth
    import ref_import as R
    cfg = cases.unet_cfg(cases.SMALL)
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    unet, refnet = build_models(cases.SMALL, synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg))
    vae = AutoencoderKLDecoder(vcfg).to("cuda")
    vae.load_state_dict(synth.vae_decoder_state_dict(vcfg))
    pipe = VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet,
                            scheduler=DDIMScheduler(**R.NOISE_SCHEDULER_KWARGS))
    Fn, cf, co, steps = 14, 4, 2, 2                       # 6 windows
    inp = synth.synthetic_inputs(cfg, Fn, 8, 8)
    outs = {}
    for upc in (2, 4, 6):
        pipe.units_per_call = upc
        outs[upc] = pipe(None, None, None, 64, 64, Fn, steps, cases.GUIDANCE, context_frames=cf, context_overlap=co,
                         reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD,
                         reference_latents=inp["ref_latents"], kps_features=inp["kps_features"],
                         audio_embeddings=inp["audio_embeddings"], latents=inp["latents"], decode=False).cpu()
    assert torch.isfinite(outs[2]).all()
    assert torch.equal(outs[2], outs[4]) and torch.equal(outs[2], outs[6])


def test_config3_window_geometry_at_sd15_widths_vs_oracle():
    """BASELINE.json configs[2] geometry - 64 frames, context 16 / overlap 4 -> 5 windows [0, 12, 24, 36, 48] - with the
    SD-1.5 widths (head dims 40 / 80 / 160, 16-frame temporal attention, merged UNet calls) on 8x8 latents, 2 DDIM
    steps, against the fp32 oracle's mean-overlap loop (oracle/loop.py: pipelines/v_express_pipeline.py:498-500,526-583)
    over the oracle UNet.  Tolerance of the loop tests: relative L2 <= 5e-2, cosine >= 0.998."""
    _need_gpu()
    from oracle import loop as OL, unet as OU
    from v_express_amd import AutoencoderKLDecoder, DDIMScheduler, VExpressPipeline, synth
    import ref_import as R
    kw, Fn, cf, co, steps = cases.FULL, 64, 16, 4, 2
    cfg, ocfg = cases.unet_cfg(kw), cases.oracle_cfg(kw)
    sd3, sd2 = synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg)
    unet, refnet = build_models(kw, sd3, sd2)
    vae = AutoencoderKLDecoder(synth.VaeConfig(**cases.SMALL_VAE)).to("cuda")
    vae.load_state_dict(synth.vae_decoder_state_dict(synth.VaeConfig(**cases.SMALL_VAE)))
    pipe = VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet,
                            scheduler=DDIMScheduler(**R.NOISE_SCHEDULER_KWARGS))
    inp = synth.synthetic_inputs(cfg, Fn, 8, 8)
    assert OL.uniform_windows(Fn, cf, co) == [list(range(s, s + 16)) for s in (0, 12, 24, 36, 48)]
    got = pipe(None, None, None, 64, 64, Fn, steps, cases.GUIDANCE, context_frames=cf, context_overlap=co,
               reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD,
               reference_latents=inp["ref_latents"], kps_features=inp["kps_features"],
               audio_embeddings=inp["audio_embeddings"], latents=inp["latents"], decode=False).cpu()
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    try:
        with torch.no_grad():
            banks = OU.reader_banks(OU.refnet_banks(sd2, ocfg, inp["ref_latents"]))
            ddim = OL.DDIM()
            ref = OL.mean_overlap(lambda x, t, e, k: OU.unet3d_forward(sd3, ocfg, x, t, e, k, banks, cases.W_REF,
                                                                       cases.W_AUD),
                                  inp["latents"], ddim.set_timesteps(steps), ddim, OL.uniform_windows(Fn, cf, co),
                                  cases.GUIDANCE, inp["kps_features"], inp["audio_embeddings"])
    finally:
        torch.set_num_threads(nthreads)
    r, c = rel_l2(got, ref), cosine(got, ref)
    print(f"[config-3 geometry, SD-1.5 widths, F=64 16/4, {steps} steps] relL2={r:.4g} cosine={c:.6f}")
    assert torch.isfinite(got).all() and r <= 5e-2 and c >= 0.998, (r, c)


def test_pipeline_without_cfg_vs_reference_golden():
    """guidance_scale = 1.0 through VExpressPipeline.__call__: no classifier-free guidance, batch of 1, the reference
    bank without the zero half (pipelines/v_express_pipeline.py:443,539,548; mutual_self_attention.py:357-363)."""
    _need_gpu()
    from v_express_amd import AutoencoderKLDecoder, DDIMScheduler, VExpressPipeline, synth
    import ref_import as R
    name, Fn, cf, co, steps = cases.NOCFG_CASE
    cfg = cases.unet_cfg(cases.SMALL)
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    unet, refnet = build_models(cases.SMALL, synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg))
    vae = AutoencoderKLDecoder(vcfg).to("cuda")
    vae.load_state_dict(synth.vae_decoder_state_dict(vcfg))
    pipe = VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet,
                            scheduler=DDIMScheduler(**R.NOISE_SCHEDULER_KWARGS))
    inp = cases.cond_only(synth.synthetic_inputs(cfg, Fn, 8, 8))
    trace = []
    video = pipe(None, None, None, 64, 64, Fn, steps, 1.0, context_frames=cf, context_overlap=co,
                 reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD,
                 reference_latents=inp["ref_latents"], kps_features=inp["kps_features"],
                 audio_embeddings=inp["audio_embeddings"], latents=inp["latents"],
                 callback=lambda i, t, l: trace.append(l.detach().cpu().clone()))
    g = torch.load(os.path.join(GOLD, f"pipeline_{name}.pt"), weights_only=False)
    r1, c1 = rel_l2(trace[-1], g["latents"]), cosine(trace[-1], g["latents"])
    mae = (video - g["video_f16"].float()).abs().mean().item()
    print(f"[{name}] final relL2={r1:.4g} cos={c1:.6f} video MAE={mae:.4g}")
    assert r1 <= 5e-2 and c1 >= 0.998 and mae <= 2e-2, (r1, c1, mae)


def test_vae_decode_vs_oracle_full_width():
    """sd-vae-ft-mse widths (128/256/512/512, attention head dim 512) on a 16x16 latent."""
    _need_gpu()
    import oracle
    from oracle import vae as OV
    from v_express_amd import AutoencoderKLDecoder, synth
    vcfg = synth.VaeConfig()
    sdv = synth.vae_decoder_state_dict(vcfg)
    vae = AutoencoderKLDecoder(vcfg).to("cuda")
    vae.load_state_dict(sdv)
    z = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    got = vae.decode(z).sample.cpu()
    ref = OV.vae_decode(sdv, oracle.VaeConfig(), z)
    r, c = rel_l2(got, ref), cosine(got, ref)
    print(f"[vae] relL2={r:.4g} cosine={c:.6f}")
    assert got.shape == (2, 3, 128, 128)
    assert r <= 3e-2 and c >= 0.999, (r, c)


def test_full_size_forward_ring_vs_classic_and_batch_invariance():
    """BASELINE.json configs[1] geometry (512^2 -> 64x64 latents, 16-frame window, SD-1.5 widths, CFG batch of 2): the
    sizes at which the persistent ring GEMM, the split-K path and attn2 actually run.  Size-independent properties:
      * the forward with the ring kernel enabled agrees with the forward on the classic tiles only (same math,
        different fp32 summation order, so bf16 roundings flip through ~200 layers: the same bound as one forward vs the
        fp32 oracle, relative L2 <= 3e-2 and cosine >= 0.999; measured 1.1e-2 / 0.99995), both finite;
      * the cond half computed alone is BIT-identical to its rows in the batched CFG call (kernel selection, K order
        and split-K factors never depend on the batch), which is what makes the multi-GPU sharding exact."""
    _need_gpu()
    import v_express_amd as vx
    from v_express_amd import lib as L, synth
    cfg = synth.UNetConfig()
    dev = torch.device("cuda")
    unet = vx.UNet3DConditionModel(cfg).to(dev)
    refnet = vx.UNet2DConditionModel(cfg).to(dev)
    unet.load_state_dict(synth.unet3d_state_dict(cfg, seed=42, device=dev, dtype=torch.bfloat16, draw_on_device=True))
    refnet.load_state_dict(synth.refnet_state_dict(cfg, seed=43, device=dev, dtype=torch.bfloat16, draw_on_device=True))
    unet.release_raw_weights()
    refnet.release_raw_weights()
    F, h, w = 16, 64, 64
    inp = synth.synthetic_inputs(cfg, F, h, w, seed=42, device=dev)
    writer = vx.ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    reader = vx.ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                          reference_attention_weight=0.95, audio_attention_weight=3.0)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768, device=dev), return_dict=False)
    reader.update(writer, True)
    x = inp["latents"].repeat(2, 1, 1, 1, 1)
    ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
    kps = inp["kps_features"]

    from v_express_amd import ops
    c0 = cfg.block_out_channels[0]

    def fwd(xx, ee, kk, rows=None):
        b = xx.shape[0]
        out = unet.forward_tokens(ops.ncfhw_to_nhwc(xx, 8), 519, ee.to(torch.bfloat16).reshape(-1, 768).contiguous(),
                                  ops.ncfhw_to_nhwc(kk, c0), b=b, f=F, H=h, W=w, batch_rows=rows)
        return out.view(b, F * h * w, -1)[..., :4].clone()
    saved = ops.RING_MODE[0]                 # (a per-call choice since ABI 14: ops turns the knob into ring_hint values)
    try:
        ops.RING_MODE[0] = 2
        ring = fwd(x, ehs, kps)
        ops.RING_MODE[0] = 0
        classic = fwd(x, ehs, kps)
    finally:
        ops.RING_MODE[0] = saved
    assert torch.isfinite(ring).all() and torch.isfinite(classic).all()
    assert rel_l2(ring, classic) <= 3e-2 and cosine(ring, classic) >= 0.999, (rel_l2(ring, classic), cosine(ring, classic))
    again = fwd(x, ehs, kps)
    assert torch.equal(again, ring)                                    # deterministic
    cond = fwd(x[1:].contiguous(), ehs[F:].contiguous(), kps[1:].contiguous(), rows=[1])
    unc = fwd(x[:1].contiguous(), ehs[:F].contiguous(), kps[:1].contiguous(), rows=[0])
    assert torch.equal(cond[0], ring[1]) and torch.equal(unc[0], ring[0])


def test_zero_audio_rows_reduce_to_output_bias():
    """The unconditional CFG half carries all-zero audio tokens (pipelines/v_express_pipeline.py:403-405): K = V = 0,
    uniform softmax, weighted sum exactly 0 -> attn2 contributes exactly w_aud * to_out.bias.  The shortcut path
    (audio_zero=[True, False], precomputed audio K|V) must reproduce the fully computed forward: the cond half
    bit-for-bit; the uncond half up to where the residual stream is rounded to bf16 - round 3: the shortcut adds
    w_ref * attn1_5 bias + w_aud * attn2 bias in the epilogue of the attn1 out-projection (ONE store of h), the full
    path adds the attn2 term in its own GEMM (a second store), i.e. one bf16 rounding of h per block apart; on this
    small, unnormalised-width model that is 1.1e-2 relative L2 at the output, reproduced to three digits by the float64
    emulation of the same host code (tests/fake_ops.py), so it is rounding placement, not arithmetic."""
    _need_gpu()
    from v_express_amd import ReferenceAttentionControl, ops, synth
    kw, F, h, w, t = cases.FORWARD_CASES["small_f8_16x8"]
    cfg = cases.unet_cfg(kw)
    sd3, sd2 = synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg)
    inp = synth.synthetic_inputs(cfg, F, h, w)
    unet, refnet = build_models(kw, sd3, sd2)
    writer = ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    reader = ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                       reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
    reader.update(writer, True)
    x = ops.ncfhw_to_nhwc(inp["latents"].repeat(2, 1, 1, 1, 1).cuda(), 8)
    ehs = inp["audio_embeddings"].reshape(-1, 768).cuda().to(torch.bfloat16).contiguous()
    assert (ehs[:F * 5] == 0).all()
    kps = ops.ncfhw_to_nhwc(inp["kps_features"].cuda(), cfg.block_out_channels[0])
    full = unet.forward_tokens(x, t, ehs, kps, b=2, f=F, H=h, W=w).view(2, F * h * w, -1)
    fast = unet.forward_tokens(x, t, ehs, kps, b=2, f=F, H=h, W=w, audio_kv=unet.precompute_audio_kv(ehs),
                               audio_zero=[True, False]).view(2, F * h * w, -1)
    assert torch.equal(fast[1], full[1])
    r, c = rel_l2(fast[0], full[0]), cosine(fast[0], full[0])
    print(f"[zero-audio shortcut] uncond half vs fully computed: relL2={r:.4g} cosine={c:.6f}")
    assert r <= 2.5e-2 and c >= 0.9995, (r, c)
    # The tight check of the shortcut's ARITHMETIC (a wrong or missing w_aud * bias term would pass the bound above): with
    # the constant added by vx_add_row_bias behind the reference attention - the rounding placement of the fully computed
    # block - the uncond half must agree with it to the round-2 bound, 12x tighter.
    ops.FOLD_ZERO_AUDIO[0] = False
    try:
        same_place = unet.forward_tokens(x, t, ehs, kps, b=2, f=F, H=h, W=w, audio_kv=unet.precompute_audio_kv(ehs),
                                         audio_zero=[True, False]).view(2, F * h * w, -1)
    finally:
        ops.FOLD_ZERO_AUDIO[0] = True
    assert torch.equal(same_place[1], full[1])
    r2 = rel_l2(same_place[0], full[0])
    print(f"[zero-audio shortcut] same rounding placement: relL2={r2:.4g}")
    assert r2 <= 2e-3, r2


def test_bench_two_rank_control_flow_on_one_gpu():
    """bench.py under torchrun with 2 ranks folded onto this GPU (VX_DIST_BACKEND=gloo: collectives staged through the
    host) - exercises exactly the multi-rank control flow the driver launches with RCCL: unit sharding, the per-step
    all-gather, the decode split + frame gather, the max-over-ranks timing and the rank-0-only roofline leg (which must
    not leave rank 0 alone inside a collective).  Checks that it terminates and prints one well-formed JSON line."""
    _need_gpu()
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VX_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    # started the way the driver may start it - plain `python bench.py --gpus 2`: bench.py re-launches itself under
    # torch.distributed.run with one rank per --gpus
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--ddim-steps", "1", "--no-cpu-baseline", "--scaling", "weak"]
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["windows"] == 2 and d["config"]["frames"] == 28
    assert d["value"] > 0 and d["scaling"] == "weak" and d["roofline"]["achieved"] > 0
    # the same clip on one rank alone, for a like-for-like multi-GPU ratio
    assert d["same_clip_1gpu_fps"] > 0 and d["speedup_vs_1gpu_same_clip"] > 0


@pytest.mark.parametrize("world,S,F,cf,co", [(2, 2, 8, 8, 2), (4, 0, 8, 8, 2), (4, 2, 14, 8, 2), (4, 1, 32, 8, 2)])
def test_frame_sharded_loop_matches_single_rank(world, S, F, cf, co, tmp_path):
    """SURVEY.md §8f rank 1 on the real kernels: S ranks per (window, CFG-half) unit, each running the UNet on f/S
    frames with the motion modules exchanging layouts through all-to-alls inside the unit's process group (ranks folded
    onto this GPU, gloo + host staging).  Against the single-rank loop on the same inputs.  Every kernel is
    row-independent, so the only differences can come from tile-configuration choices that look at the per-rank row
    count: bf16-rounding level (<= 1e-2 relative L2), usually exactly zero.  S = 0: the automatic policy (4 ranks,
    one window -> 2 ranks per unit)."""
    _need_gpu()
    import subprocess
    import sys
    import dist_gpu_worker as W
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "lat.pt")
    env = dict(os.environ, VX_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29540 + 10 * world + S + F),
           os.path.join(root, "tests", "dist_gpu_worker.py"), out, str(F), str(cf), str(co), "2", str(S)]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = torch.load(out)
    ref = W.run(F, cf, co, 2, 0)                        # this process: no process group -> single rank
    err = rel_l2(got, ref)
    print(f"[frame shards world={world} S={S} F={F}] relL2 vs single rank = {err:.3g}, "
          f"bit-identical = {torch.equal(got, ref)}")
    assert torch.isfinite(got).all() and err <= 1e-2, err
