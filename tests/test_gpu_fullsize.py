"""BASELINE.json configs[1] - the benchmarked configuration itself - against outputs of the REFERENCE run at that size
(tests/golden/fullsize_F16_512.pt, written by `tests/make_golden.py fullsize`: the reference's own modules, its own
`VExpressPipeline.__call__` loop (pipelines/v_express_pipeline.py:526-589) and `decode_latents` (:152-166), fp32 on the
host cores, SD-1.5 widths, 64x64 latents = 512x512, one 16-frame window, CFG 3.5, 25 DDIM steps, sd-vae-ft-mse-shaped
VAE, seeded synthetic weights and inputs of v_express_amd.synth).

Stated tolerances (bf16 storage + fp32 accumulation vs the fp32 reference):
  one 16-frame CFG UNet3D forward (the first of the loop, t = 999)     relative L2 <= 3e-2, cosine >= 0.999
  latents after DDIM steps 1 / 5 / 13 / 25                              relative L2 <= 3e-3 / 1e-2 / 2e-2 / 3e-2
                                                                        (measured 2.7e-4 / 1.0e-3 / 3.4e-3 / 7.2e-3)
  decoded 512x512 frames of the 25-step clip                            PSNR >= 40 dB (measured 50.8 dB)
  VAE decode alone, of the REFERENCE's final latents                    PSNR >= 35 dB, mean abs error <= 1e-2
  fp8 projections (BASELINE configs[4], `unet.fp8_projections = True`; no reference counterpart - the reference's
  fp32 goldens are the yardstick): first forward relative L2 <= 6e-2, cosine >= 0.998; 25-step latents relative L2
  <= 5e-3 / 2e-2 / 4e-2 / 6e-2 after steps 1 / 5 / 13 / 25; decoded frames PSNR >= 35 dB
The relative-L2 bounds are the ones with teeth: the start noise itself has cosine 0.99994 / 0.9968 / 0.951 / 0.746 with
the reference's latents after steps 1 / 5 / 13 / 25, i.e. relative L2 1.1e-2 / 8e-2 / 0.31 / 0.67 - a loop that does
nothing fails every one of them (a cosine >= 0.99 bound does not, for steps 1 and 5).
"""
import os

import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_F16_512.pt")


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).sum().sqrt() / (b.pow(2).sum().sqrt() + 1e-12)).item()


def cosine(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-12)).item()


def psnr(a, b):
    mse = (a.float().cpu() - b.float().cpu()).pow(2).mean().item()
    return 10 * torch.log10(torch.tensor(1.0 / max(mse, 1e-12))).item()


@pytest.fixture(scope="module")
def full():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if not os.path.exists(GOLD):
        pytest.fail("tests/golden/fullsize_F16_512.pt is missing (python tests/make_golden.py fullsize)")
    from v_express_amd import (AutoencoderKLDecoder, DDIMScheduler, UNet2DConditionModel, UNet3DConditionModel,
                               VExpressPipeline, synth)
    import ref_import as R
    cfg, vcfg = cases.unet_cfg(cases.FULL), synth.VaeConfig()
    unet = UNet3DConditionModel(cfg).to("cuda")
    refnet = UNet2DConditionModel(cfg).to("cuda")
    # the CPU draw (the reference ran on exactly these tensors), then the device layouts; the source copies are dropped
    unet.load_state_dict(synth.unet3d_state_dict(cfg), strict=True)
    unet.release_raw_weights()
    refnet.load_state_dict(synth.refnet_state_dict(cfg), strict=True)
    refnet.release_raw_weights()
    vae = AutoencoderKLDecoder(vcfg).to("cuda")
    vae.load_state_dict(synth.vae_decoder_state_dict(vcfg))
    pipe = VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet,
                            scheduler=DDIMScheduler(**R.NOISE_SCHEDULER_KWARGS))
    F, cf, co, steps = cases.FULLSIZE_CASE
    return dict(pipe=pipe, cfg=cfg, inp=synth.synthetic_inputs(cfg, F, 64, 64),
                gold=torch.load(GOLD, weights_only=False))


def test_fullsize_16_frame_forward_vs_reference_golden(full):
    """The first UNet call of the reference loop: sample = the start latents twice (CFG), t = 999, 16 frames."""
    from v_express_amd import ReferenceAttentionControl
    pipe, inp, g = full["pipe"], full["inp"], full["gold"]
    unet, refnet = pipe.denoising_unet, pipe.reference_net
    writer = ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    reader = ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                       reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
    reader.update(writer, True)
    x = inp["latents"].repeat(2, 1, 1, 1, 1)
    ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
    got = unet(x, 999, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
    reader.clear()
    writer.clear()
    want = g["pred_step0"]
    r, c = rel_l2(got, want), cosine(got, want)
    print(f"[fullsize f=16 forward, t=999] relL2={r:.4g} cosine={c:.6f}")
    assert got.shape == want.shape == (2, 4, 16, 64, 64) and torch.isfinite(got).all()
    assert r <= 3e-2 and c >= 0.999, (r, c)


LOOP_STEPS = ((0, "latents_step0"), (4, "latents_step4"), (12, "latents_step12"), (24, "latents"))
LOOP_BOUNDS_BF16 = {0: 3e-3, 4: 1e-2, 12: 2e-2, 24: 3e-2}      # relative L2 of the latents after step i + 1
LOOP_BOUNDS_FP8 = {0: 5e-3, 4: 2e-2, 12: 4e-2, 24: 6e-2}


def _check_loop(trace, video, g, tag, bounds, min_psnr):
    """Latents after steps 1 / 5 / 13 / 25 and the decoded frames against the reference's; also proves that the bounds
    reject a loop that never moves the latents (the start noise against the same goldens)."""
    bad = []
    for i, key in LOOP_STEPS:
        r, c = rel_l2(trace[i], g[key]), cosine(trace[i], g[key])
        print(f"[fullsize loop {tag}] after step {i + 1:2d}: relL2={r:.4g} cosine={c:.6f}  (bound {bounds[i]:.0e})")
        if not r <= bounds[i]:
            bad.append((i + 1, r, bounds[i]))
    frames = list(g["video_frames"])
    p = psnr(video[:, :, frames], g["video_f16"])
    mae = (video[:, :, frames] - g["video_f16"].float()).abs().mean().item()
    print(f"[fullsize loop {tag}] decoded frames {frames}: PSNR={p:.1f} dB, MAE={mae:.4g}  (bound {min_psnr} dB)")
    assert torch.isfinite(video).all() and 0.0 <= video.min().item() and video.max().item() <= 1.0
    assert not bad, f"latents left the stated relative-L2 bound: {bad}"
    assert p >= min_psnr, f"decoded-frame PSNR {p:.1f} dB"


def test_loop_bounds_reject_a_loop_that_does_nothing(full):
    """The untouched start noise must FAIL every latent bound (the round-2 cosine >= 0.99 bound did not at steps 1, 5)."""
    g, noise = full["gold"], full["inp"]["latents"]
    for i, key in LOOP_STEPS:
        r = rel_l2(noise, g[key])
        print(f"[fullsize loop] start noise vs reference after step {i + 1:2d}: relL2={r:.4g}")
        assert r > 2 * LOOP_BOUNDS_FP8[i] and r > 3 * LOOP_BOUNDS_BF16[i], (i, r)


def test_fullsize_25_step_call_vs_reference_golden(full):
    """`VExpressPipeline.__call__`: 25 DDIM steps + decode, against the reference's latents along the way and its
    decoded frames."""
    pipe, inp, g = full["pipe"], full["inp"], full["gold"]
    F, cf, co, steps = cases.FULLSIZE_CASE
    trace = {}
    video = pipe(None, None, None, 512, 512, F, steps, cases.GUIDANCE, context_frames=cf, context_overlap=co,
                 reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD,
                 reference_latents=inp["ref_latents"], kps_features=inp["kps_features"],
                 audio_embeddings=inp["audio_embeddings"], latents=inp["latents"],
                 callback=lambda i, t, l: trace.__setitem__(i, l.detach().cpu().clone()) if i in (0, 4, 12, 24) else None)
    assert video.shape == (1, 3, F, 512, 512) and video.dtype == torch.float32 and video.device.type == "cpu"
    _check_loop(trace, video, g, "bf16", LOOP_BOUNDS_BF16, 40.0)


def test_fullsize_fp8_projections_forward_and_25_step_call_vs_reference_golden(full):
    """BASELINE.json configs[4]'s fp8 q / k / v / out projections at the headline size: the first 16-frame CFG forward
    and the whole 25-step call + decode, against the REFERENCE's fp32 goldens (the reference has no fp8 mode; stated
    tolerances in the module docstring).  pipelines/v_express_pipeline.py:526-589."""
    from v_express_amd import ReferenceAttentionControl
    pipe, inp, g = full["pipe"], full["inp"], full["gold"]
    unet, refnet = pipe.denoising_unet, pipe.reference_net
    F, cf, co, steps = cases.FULLSIZE_CASE
    unet.fp8_projections = True
    try:
        writer = ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
        reader = ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                           reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD)
        refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
        reader.update(writer, True)
        x = inp["latents"].repeat(2, 1, 1, 1, 1)
        ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
        got = unet(x, 999, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
        reader.clear()
        writer.clear()
        r, c = rel_l2(got, g["pred_step0"]), cosine(got, g["pred_step0"])
        print(f"[fullsize f=16 forward, fp8 projections] relL2={r:.4g} cosine={c:.6f}")
        assert torch.isfinite(got).all() and r <= 6e-2 and c >= 0.998, (r, c)
        trace = {}
        video = pipe(None, None, None, 512, 512, F, steps, cases.GUIDANCE, context_frames=cf, context_overlap=co,
                     reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD,
                     reference_latents=inp["ref_latents"], kps_features=inp["kps_features"],
                     audio_embeddings=inp["audio_embeddings"], latents=inp["latents"],
                     callback=lambda i, t, l: trace.__setitem__(i, l.detach().cpu().clone())
                     if i in (0, 4, 12, 24) else None)
    finally:
        unet.fp8_projections = False
    _check_loop(trace, video, g, "fp8", LOOP_BOUNDS_FP8, 35.0)


def test_fullsize_vae_decode_512_vs_reference_golden(full):
    """AutoencoderKLDecoder at a 64x64 latent (M = 262144 output rows per frame at the last level): the reference's
    final latents through OUR decode_latents against the reference's decoded frames."""
    pipe, g = full["pipe"], full["gold"]
    frames = list(g["video_frames"])
    lat = g["latents"][:, :, frames].to("cuda")
    video = pipe.decode_latents(lat).cpu()
    p = psnr(video, g["video_f16"])
    mae = (video - g["video_f16"].float()).abs().mean().item()
    print(f"[fullsize VAE decode 512x512] PSNR={p:.1f} dB, MAE={mae:.4g}")
    assert video.shape == (1, 3, len(frames), 512, 512)
    assert p >= 35.0 and mae <= 1e-2, (p, mae)


# ---------------------------------------------------------------------------------------------------------------------
# The sliding-window / merged-call path at the BENCHMARKED geometry (BASELINE.json configs[2] / [3] report it):
# tests/golden/fullsize_F28_512.pt = the reference's own `VExpressPipeline.__call__` at SD-1.5 widths, 64x64 latents,
# F = 28, context 16 / overlap 4 (windows [0..15] and [12..27], four frames averaged), CFG 3.5, 2 DDIM steps.
GOLD_F28 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_F28_512.pt")
F28_BOUNDS = (1.5e-2, 3e-2)        # relative L2 of the latents after step 1 / step 2 (two LARGE steps: t = 999, 499)


def _run_f28(pipe, inp, units_per_call, spy=None, case=None):
    F, cf, co, steps = case or cases.FULLSIZE_F28_CASE
    unet = pipe.denoising_unet
    saved_upc, saved_fwd = pipe.units_per_call, unet.forward_tokens
    trace = {}
    if spy is not None:
        def wrapped(*a, **k):
            o = saved_fwd(*a, **k)
            spy.append((k["b"], o.detach().clone()))
            return o
        unet.forward_tokens = wrapped
    pipe.units_per_call = units_per_call
    try:
        lat = pipe(None, None, None, 512, 512, F, steps, cases.GUIDANCE, context_frames=cf, context_overlap=co,
                   reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD,
                   reference_latents=inp["ref_latents"], kps_features=inp["kps_features"],
                   audio_embeddings=inp["audio_embeddings"], latents=inp["latents"], decode=False,
                   callback=lambda i, t, l: trace.__setitem__(i, l.detach().clone()))
    finally:
        pipe.units_per_call = saved_upc
        unet.forward_tokens = saved_fwd
    return lat, trace


def test_fullsize_two_overlapping_windows_one_merged_call_vs_reference_golden(full):
    """b = 4 UNet calls (both CFG halves of both windows in ONE launch sequence: M = 262 144 rows at the 64x64 level) and
    the overlap averaging of frames 12-15 at 64x64 latents: every window's first prediction and the latents after each of
    the two steps against the reference's own run; then the same clip with units_per_call = 2 (one call per window) must
    give the same BITS (the kernels are batch-invariant).  pipelines/v_express_pipeline.py:526-589."""
    if not os.path.exists(GOLD_F28):
        pytest.fail("tests/golden/fullsize_F28_512.pt is missing (python tests/make_golden.py fullsize_F28)")
    from v_express_amd import synth
    pipe, cfg = full["pipe"], full["cfg"]
    g = torch.load(GOLD_F28, weights_only=False)
    F, cf, co, steps = cases.FULLSIZE_F28_CASE
    inp = synth.synthetic_inputs(cfg, F, 64, 64)
    calls = []
    lat4, trace4 = _run_f28(pipe, inp, 4, spy=calls)
    assert [b for b, _ in calls] == [4] * steps, [b for b, _ in calls]            # one merged call per step
    first = calls[0][1].view(4, 16, 64 * 64, -1)[..., :4].permute(0, 3, 1, 2).reshape(4, 4, 16, 64, 64)
    want = g["pred_step0_f16"].float()
    for wi in range(2):                      # rows: (window 0: uncond, cond), (window 1: uncond, cond) - the reference's order
        r, c = rel_l2(first[2 * wi:2 * wi + 2], want[2 * wi:2 * wi + 2]), cosine(first[2 * wi:2 * wi + 2], want[2 * wi:2 * wi + 2])
        print(f"[F28 merged call] window {wi} first prediction: relL2={r:.4g} cosine={c:.6f}")
        assert r <= 3e-2 and c >= 0.999, (wi, r, c)
    for i, key in ((0, "latents_step0"), (1, "latents")):
        r, rn = rel_l2(trace4[i], g[key]), rel_l2(inp["latents"], g[key])
        print(f"[F28 merged call] latents after step {i + 1}: relL2={r:.4g} (bound {F28_BOUNDS[i]:.1e}; untouched noise {rn:.3g})")
        assert r <= F28_BOUNDS[i], (i, r)
        assert rn > 3 * F28_BOUNDS[i], (i, rn)                                    # the bound rejects a loop that does nothing
    # one call per window (b = 2) instead of the merged b = 4 call: same bits
    lat2, trace2 = _run_f28(pipe, inp, 2, spy=(calls2 := []))
    assert [b for b, _ in calls2] == [2] * (2 * steps)
    # (diagnostic first: the raw UNet outputs of step 1, window by window, then the latents)
    merged = calls[0][1].view(4, -1)
    for wi in range(2):
        sep = calls2[wi][1].view(2, -1)
        same = torch.equal(sep, merged[2 * wi:2 * wi + 2])
        print(f"[F28] step 1, window {wi}: b = 2 call vs rows of the b = 4 call: {'bit-identical' if same else 'DIFFERENT'}"
              f" (max |diff| {(sep - merged[2 * wi:2 * wi + 2]).abs().max().item():.3g})")
        assert same, f"window {wi}: the merged b = 4 call is not bit-identical to the b = 2 call"
    assert torch.equal(lat2, lat4) and torch.equal(trace2[0], trace4[0])


GOLD_CTX24 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_ctx24_F44_512.pt")


def test_fullsize_reference_default_window_24_overlap_4_vs_reference_golden(full):
    """The reference's DEFAULT window geometry (inference.py:67-68: --context_frames 24 --context_overlap 4; the motion
    modules' positional table holds 32 entries, inference_v2.yaml:21) at SD-1.5 widths and 64x64 latents: F = 44 = windows
    [0..23] and [20..43] sharing four frames, CFG 3.5, 2 DDIM steps, against the reference's own
    `VExpressPipeline.__call__` (tests/golden/fullsize_ctx24_F44_512.pt, make_golden.py fullsize_ctx24).  Every model-level
    test before round 5 used windows of 4, 8 or 16 frames; f = 24 exercises the temporal attention
    past 16 frames (modules/motion_module.py:236-259,351-388) in every motion module of the UNet.  Also prints which
    implementation the temporal blocks took at this window length (ops.block_paths)."""
    if not os.path.exists(GOLD_CTX24):
        pytest.fail("tests/golden/fullsize_ctx24_F44_512.pt is missing (python tests/make_golden.py fullsize_ctx24)")
    from v_express_amd import ops, synth
    pipe, cfg = full["pipe"], full["cfg"]
    g = torch.load(GOLD_CTX24, weights_only=False)
    F, cf, co, steps = cases.FULLSIZE_CTX24_CASE
    assert [list(w) for w in g["windows"]] == [list(range(0, 24)), list(range(20, 44))]
    inp = synth.synthetic_inputs(cfg, F, 64, 64)
    calls = []
    lat, trace = _run_f28(pipe, inp, 2, spy=calls, case=cases.FULLSIZE_CTX24_CASE)      # one b = 2 call per window
    assert [b for b, _ in calls] == [2] * (2 * steps), [b for b, _ in calls]
    want = g["pred_step0_f16"].float()
    assert tuple(want.shape) == (4, 4, cf, 64, 64)
    for wi in range(2):                      # rows: (window wi: uncond, cond) - the reference's order
        first = calls[wi][1].view(2, cf, 64 * 64, -1)[..., :4].permute(0, 3, 1, 2).reshape(2, 4, cf, 64, 64)
        r, c = rel_l2(first, want[2 * wi:2 * wi + 2]), cosine(first, want[2 * wi:2 * wi + 2])
        print(f"[ctx 24 / overlap 4] window {wi} first prediction: relL2={r:.4g} cosine={c:.6f}")
        assert r <= 3e-2 and c >= 0.999, (wi, r, c)
    for i, key in ((0, "latents_step0"), (1, "latents")):
        r, rn = rel_l2(trace[i], g[key]), rel_l2(inp["latents"], g[key])
        print(f"[ctx 24 / overlap 4] latents after step {i + 1}: relL2={r:.4g} (bound {F28_BOUNDS[i]:.1e}; untouched noise {rn:.3g})")
        assert r <= F28_BOUNDS[i], (i, r)
        assert rn > 3 * F28_BOUNDS[i], (i, rn)                                    # the bound rejects a loop that does nothing
    paths = {k: v for k, v in ops.block_paths().items() if k.startswith("temporal_attention") and " f=24 " in k}
    print("[ctx 24 / overlap 4] temporal blocks:", paths)
    assert paths, "no temporal-attention path was recorded for f = 24"


GOLD_768 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_768_F4.pt")


def test_768_loop_and_decode_at_96x96_latents(full):
    """BASELINE.json configs[4]'s geometry: a 2-step, 4-frame CFG loop at 96x96 latents against the reference's own run
    (tests/golden/fullsize_768_F4.pt, make_golden.py fullsize_768), and `decode_latents` of one 768x768 frame against the
    fp32 oracle (oracle/vae.py, restating the sd-vae-ft-mse decoder the reference calls at
    pipelines/v_express_pipeline.py:152-166)."""
    if not os.path.exists(GOLD_768):
        pytest.fail("tests/golden/fullsize_768_F4.pt is missing (python tests/make_golden.py fullsize_768)")
    import oracle
    from oracle import vae as OV
    from v_express_amd import synth
    pipe, cfg = full["pipe"], full["cfg"]
    g = torch.load(GOLD_768, weights_only=False)
    F, cf, co, steps = cases.FULLSIZE_768_CASE
    inp = synth.synthetic_inputs(cfg, F, 96, 96)
    trace = {}
    lat = pipe(None, None, None, 768, 768, F, steps, cases.GUIDANCE, context_frames=cf, context_overlap=co,
               reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD,
               reference_latents=inp["ref_latents"], kps_features=inp["kps_features"],
               audio_embeddings=inp["audio_embeddings"], latents=inp["latents"], decode=False,
               callback=lambda i, t, l: trace.__setitem__(i, l.detach().clone()))
    for i, key in ((0, "latents_step0"), (1, "latents")):
        r, rn = rel_l2(trace[i], g[key]), rel_l2(inp["latents"], g[key])
        print(f"[768x768 loop] latents after step {i + 1}: relL2={r:.4g} (bound {F28_BOUNDS[i]:.1e}; untouched noise {rn:.3g})")
        assert r <= F28_BOUNDS[i] and rn > 3 * F28_BOUNDS[i], (i, r, rn)
    # decode of ONE 768x768 frame (the reference's final latents) vs the oracle on the host cores (~10-20 s)
    vcfg = synth.VaeConfig()
    z = g["latents"][:, :, :1]
    video = pipe.decode_latents(z.to("cuda")).cpu()
    sdv = synth.vae_decoder_state_dict(vcfg)
    with torch.no_grad():
        ref = OV.decode_latents(sdv, oracle.VaeConfig(), z)                 # [1, 3, 1, 768, 768] in [0, 1]
    p = psnr(video, ref)
    print(f"[768x768 VAE decode] PSNR={p:.1f} dB")
    assert video.shape == (1, 3, 1, 768, 768) and p >= 35.0, p
