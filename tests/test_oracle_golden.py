"""oracle/ (Tier B) against the committed golden vectors produced by the reference itself (Tier A,
tests/make_golden.py).  CPU only; runs in the dev container and on the GPU box."""
import os

import pytest
import torch

import cases
import oracle
from oracle import loop as OL
from oracle import unet as OU
from oracle import vae as OV
from v_express_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-5          # fp32 round-off between two orderings of the same arithmetic


def _load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


@pytest.mark.parametrize("name", ["small_f4_8x8", "small_f8_16x8", "full_f4_8x8"])
def test_unet_forward_matches_reference_golden(name):
    kw, F, h, w, t = cases.FORWARD_CASES[name]
    cfg, ocfg = cases.unet_cfg(kw), cases.oracle_cfg(kw)
    sd3, sd2 = synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg)
    inp = synth.synthetic_inputs(cfg, F, h, w)
    g = _load(f"forward_{name}.pt")
    banks = OU.refnet_banks(sd2, ocfg, inp["ref_latents"])
    assert sorted(banks) == sorted(g["bank_stats"])
    for k, v in banks.items():
        st = torch.stack([v.mean(), v.abs().mean(), v.flatten()[::97].sum()])
        assert torch.allclose(st, g["bank_stats"][k], atol=1e-3, rtol=1e-4), k
    x = inp["latents"].repeat(2, 1, 1, 1, 1)
    ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
    pred = OU.unet3d_forward(sd3, ocfg, x, t, ehs, inp["kps_features"], OU.reader_banks(banks),
                             cases.W_REF, cases.W_AUD)
    assert pred.shape == g["pred"].shape
    assert (pred - g["pred"]).abs().max().item() < TOL


@pytest.mark.parametrize("name", list(cases.PIPELINE_CASES))
def test_loop_and_decode_match_reference_golden(name):
    F, cf, co, steps = cases.PIPELINE_CASES[name]
    kw = cases.SMALL
    cfg, ocfg = cases.unet_cfg(kw), cases.oracle_cfg(kw)
    vcfg, ovcfg = synth.VaeConfig(**cases.SMALL_VAE), oracle.VaeConfig(**cases.SMALL_VAE)
    sd3, sd2, sdv = synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg), synth.vae_decoder_state_dict(vcfg)
    inp = synth.synthetic_inputs(cfg, F, 8, 8)
    g = _load(f"pipeline_{name}.pt")
    banks = OU.reader_banks(OU.refnet_banks(sd2, ocfg, inp["ref_latents"]))
    ddim = OL.DDIM()
    ts = ddim.set_timesteps(steps)
    wins = OL.uniform_windows(F, cf, co)
    trace = []
    lat = OL.mean_overlap(lambda x, t, e, k: OU.unet3d_forward(sd3, ocfg, x, t, e, k, banks, cases.W_REF, cases.W_AUD),
                          inp["latents"], ts, ddim, wins, cases.GUIDANCE, inp["kps_features"],
                          inp["audio_embeddings"], callback=lambda i, t, l: trace.append(l.clone()))
    assert (trace[0] - g["latents_step0"]).abs().max().item() < TOL
    assert (lat - g["latents"]).abs().max().item() < 5 * TOL
    video = OV.decode_latents(sdv, ovcfg, lat)
    assert video.shape == g["video_f16"].shape
    assert (video - g["video_f16"].float()).abs().max().item() < 2e-3      # fp16 storage of the fixture


def test_windows_match_reference_context_py():
    for (F, cs, co), ref in _load("windows.pt").items():
        assert OL.uniform_windows(F, cs, co) == [list(map(int, r)) for r in ref], (F, cs, co)
    assert OL.uniform_windows(64, 16, 4)[1][0] == 12
    assert OL.uniform_windows(128, 16, 4)[-1] == [120, 121, 122, 123, 124, 125, 126, 127,
                                                  126, 125, 124, 123, 122, 121, 120, 119]


def test_ddim_constants_match_scheduler():
    g = _load("ddim.pt")
    d = OL.DDIM()
    assert d.set_timesteps(25) == g["timesteps"].tolist() == list(range(999, 0, -40))
    assert torch.allclose(d.alphas_cumprod, g["alphas_cumprod"], atol=1e-7)
    assert d.alphas_cumprod[999].item() == 0.0        # zero terminal SNR
    # known answers that do not pass through the diffusers stand-in: the Stable Diffusion schedule's published
    # endpoints (abar_0 = 1 - 0.00085, abar_999 = 0.00466) and the zero-terminal-SNR rescale of them in closed form
    import numpy as np
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    abar = np.cumprod(1.0 - betas)
    assert abs(abar[0] - 0.99915) < 1e-9 and abs(abar[-1] - 0.00466) < 1e-5
    r = np.sqrt(abar)
    want = ((r - r[-1]) * r[0] / (r[0] - r[-1])) ** 2
    assert np.abs(d.alphas_cumprod.double().numpy() - want).max() < 2e-6
    # one DDIM step (eta = 0, v-prediction) keeps a sample that the model predicts perfectly on its trajectory:
    # x_t = sqrt(a) x0 + sqrt(1-a) eps, v = sqrt(a) eps - sqrt(1-a) x0  ->  x_prev = sqrt(a') x0 + sqrt(1-a') eps
    gen = torch.Generator().manual_seed(3)
    x0, eps = torch.randn(64, generator=gen), torch.randn(64, generator=gen)
    for t in (959, 479, 39):
        a, ap = d.alphas_cumprod[t], (d.alphas_cumprod[t - 40] if t >= 40 else torch.tensor(1.0))
        xt = a.sqrt() * x0 + (1 - a).sqrt() * eps
        v = a.sqrt() * eps - (1 - a).sqrt() * x0
        assert torch.allclose(d.step(v, t, xt), ap.sqrt() * x0 + (1 - ap).sqrt() * eps, atol=2e-6)


def test_prologue_oracle_matches_reference_golden():
    """oracle/prologue.py against tests/golden/prologue.pt (reference outputs, tests/make_golden.py prologue)."""
    from oracle import prologue as OP
    g = _load("prologue.pt")
    inp = cases.prologue_inputs()
    b, c, f, H, W = inp["kps_images"].shape
    frames = inp["kps_images"].permute(0, 2, 1, 3, 4).reshape(b * f, c, H, W)
    for tag, kw in (("small", cases.KPS_SMALL), ("full", {})):
        sd = synth.kps_guider_state_dict(synth.KpsGuiderConfig(**kw))
        got = OP.kps_guider(sd, frames).reshape(b, f, -1, H // 8, W // 8).permute(0, 2, 1, 3, 4)
        assert (got - g[f"kps_{tag}"]).abs().max().item() < TOL
    for tag, kw, key in (("small", cases.AUDIO_SMALL, "audio_windows_small"), ("full", {}, "audio_windows_full")):
        acfg = synth.AudioProjectionConfig(**kw)
        got = OP.audio_projection(synth.audio_projection_state_dict(acfg), inp[key], acfg.depth, acfg.heads)
        assert (got - g[f"audio_{tag}"]).abs().max().item() < 2 * TOL
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    got = OP.vae_encode_mean(synth.vae_encoder_state_dict(vcfg), oracle.VaeConfig(**cases.SMALL_VAE), inp["ref_image"])
    assert (got - g["vae_mean"]).abs().max().item() < 2 * TOL
    med = OP.median_filter_3d(inp["video"], 3)
    assert torch.equal(med, g["median"])
    assert torch.equal(torch.from_numpy(OP.frames_uint8(med)), g["median_u8"])
    assert torch.equal(OP.audio_windows(inp["wav2vec_states"], 7, 2), g["audio_windows_F7"])


@pytest.mark.parametrize("tag", ["small", "base"])
def test_wav2vec2_oracle_matches_transformers_golden_and_live(tag):
    """oracle/wav2vec2.py against tests/golden/wav2vec2.pt (outputs of transformers' Wav2Vec2Model, the third-party
    module the reference calls; tests/make_golden.py wav2vec2) and, where transformers is importable, against the
    library itself on the same weights (strict state_dict load == key-schema check)."""
    from oracle import wav2vec2 as OW
    g = _load("wav2vec2.pt")
    kw, samples = cases.W2V_CASES[tag]
    cfg = synth.Wav2Vec2Config(**kw)
    sd = synth.wav2vec2_state_dict(cfg)
    wav = cases.waveform(samples)
    got = OW.forward(sd, wav, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.num_conv_pos_embedding_groups,
                     cfg.conv_stride, cfg.layer_norm_eps)
    assert got.shape == g[tag].shape and (got - g[tag]).abs().max().item() < 2 * TOL
    try:
        hf = cases.hf_wav2vec2(cfg, sd)
    except ImportError:
        return
    with torch.no_grad():
        live = hf(wav).last_hidden_state
    assert (got - live).abs().max().item() < 2 * TOL
    # legacy weight_norm key names (checkpoints saved before torch's parametrization API) give the same weight
    p = "encoder.pos_conv_embed.conv."
    legacy = {k: v for k, v in sd.items() if "parametrizations" not in k}
    legacy[p + "weight_g"] = sd[p + "parametrizations.weight.original0"]
    legacy[p + "weight_v"] = sd[p + "parametrizations.weight.original1"]
    assert torch.equal(OW.pos_conv_weight(legacy), OW.pos_conv_weight(sd))


def test_loop_without_cfg_matches_reference_golden():
    """guidance_scale = 1.0 (pipelines/v_express_pipeline.py:443: no classifier-free guidance): batch of 1, conditional
    inputs only, reference bank without the zero half (mutual_self_attention.py:357-363)."""
    name, F, cf, co, steps = cases.NOCFG_CASE
    cfg, ocfg = cases.unet_cfg(cases.SMALL), cases.oracle_cfg(cases.SMALL)
    vcfg, ovcfg = synth.VaeConfig(**cases.SMALL_VAE), oracle.VaeConfig(**cases.SMALL_VAE)
    sd3, sd2, sdv = synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg), synth.vae_decoder_state_dict(vcfg)
    inp = cases.cond_only(synth.synthetic_inputs(cfg, F, 8, 8))
    g = _load(f"pipeline_{name}.pt")
    banks = OU.reader_banks(OU.refnet_banks(sd2, ocfg, inp["ref_latents"]), do_classifier_free_guidance=False)
    ddim = OL.DDIM()
    lat = OL.mean_overlap(lambda x, t, e, k: OU.unet3d_forward(sd3, ocfg, x, t, e, k, banks, cases.W_REF, cases.W_AUD),
                          inp["latents"], ddim.set_timesteps(steps), ddim, OL.uniform_windows(F, cf, co), 1.0,
                          inp["kps_features"], inp["audio_embeddings"])
    assert (lat - g["latents"]).abs().max().item() < 5 * TOL
    video = OV.decode_latents(sdv, ovcfg, lat)
    assert (video - g["video_f16"].float()).abs().max().item() < 2e-3
