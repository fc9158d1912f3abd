"""BASELINE.json configs[1] - the benchmarked configuration itself - against outputs of the REFERENCE run at that size
(tests/golden/fullsize_F16_512.pt, written by `tests/make_golden.py fullsize`: the reference's own modules, its own
`VExpressPipeline.__call__` loop (pipelines/v_express_pipeline.py:526-589) and `decode_latents` (:152-166), fp32 on the
host cores, SD-1.5 widths, 64x64 latents = 512x512, one 16-frame window, CFG 3.5, 25 DDIM steps, sd-vae-ft-mse-shaped
VAE, seeded synthetic weights and inputs of v_express_amd.synth).

Stated tolerances (bf16 storage + fp32 accumulation vs the fp32 reference):
  one 16-frame CFG UNet3D forward (the first of the loop, t = 999)     relative L2 <= 3e-2, cosine >= 0.999
  latents after DDIM steps 1 / 5 / 13 / 25                              cosine >= 0.99 (relative L2 printed)
  decoded 512x512 frames of the 25-step clip                            PSNR >= 30 dB
  VAE decode alone, of the REFERENCE's final latents                    PSNR >= 35 dB, mean abs error <= 1e-2
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
    worst = 1.0
    for i, key in ((0, "latents_step0"), (4, "latents_step4"), (12, "latents_step12"), (24, "latents")):
        r, c = rel_l2(trace[i], g[key]), cosine(trace[i], g[key])
        worst = min(worst, c)
        print(f"[fullsize loop] after step {i + 1:2d}: relL2={r:.4g} cosine={c:.6f}")
    frames = list(g["video_frames"])
    p = psnr(video[:, :, frames], g["video_f16"])
    mae = (video[:, :, frames] - g["video_f16"].float()).abs().mean().item()
    print(f"[fullsize loop] decoded frames {frames}: PSNR={p:.1f} dB, MAE={mae:.4g}")
    assert torch.isfinite(video).all() and 0.0 <= video.min().item() and video.max().item() <= 1.0
    assert worst >= 0.99, f"latent cosine fell to {worst:.5f} along the 25 steps"
    assert p >= 30.0, f"decoded-frame PSNR {p:.1f} dB"


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
