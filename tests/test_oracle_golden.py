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
