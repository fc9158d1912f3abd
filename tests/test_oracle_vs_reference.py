"""Tier A == Tier B, live: the reference's own modules (imported unmodified) against oracle/ on fresh seeds.
Dev container only (skips where /root/reference is absent).  Also pins the synthetic state_dict schema
to the reference's own `state_dict()` keys/shapes."""
import torch

import cases
from oracle import unet as OU
from v_express_amd import synth


def test_schema_and_forward_against_live_reference(reference):
    import refharness as H
    kw = cases.SMALL
    cfg, ocfg = cases.unet_cfg(kw), cases.oracle_cfg(kw)
    sd3, sd2 = synth.unet3d_state_dict(cfg, seed=7), synth.refnet_state_dict(cfg, seed=8)
    unet, refnet = H.build_reference_unets(cfg, sd3, sd2)     # strict=True load == schema check
    r3, r2 = unet.state_dict(), refnet.state_dict()
    assert set(r3) == set(sd3) and set(r2) == set(sd2)
    assert all(tuple(r3[k].shape) == tuple(sd3[k].shape) for k in sd3)
    assert len(sd3) == 1386 and len(sd2) == 684
    inp = synth.synthetic_inputs(cfg, 6, 8, 16, seed=9)
    frames = [1, 2, 3, 5]
    pred, banks = H.reference_unet_forward(unet, refnet, inp, 519, 0.9, 2.0, frames=frames)
    ob = OU.refnet_banks(sd2, ocfg, inp["ref_latents"])
    assert max((ob[k] - banks[k]).abs().max().item() for k in ob) < 5e-5
    x = inp["latents"][:, :, frames].repeat(2, 1, 1, 1, 1)
    ehs = inp["audio_embeddings"][:, frames].reshape(-1, 5, 768)
    o = OU.unet3d_forward(sd3, ocfg, x, 519, ehs, inp["kps_features"][:, :, frames], OU.reader_banks(ob), 0.9, 2.0)
    assert (o - pred).abs().max().item() < 2e-5


def test_uncond_reference_attention_is_out_bias(reference):
    """attention against an all-zero bank returns exactly to_out.bias."""
    from oracle import leaf as L
    g = torch.Generator().manual_seed(0)
    w = {"a.to_q.weight": torch.randn(64, 64, generator=g), "a.to_k.weight": torch.randn(64, 64, generator=g),
         "a.to_v.weight": torch.randn(64, 64, generator=g), "a.to_out.0.weight": torch.randn(64, 64, generator=g),
         "a.to_out.0.bias": torch.randn(64, generator=g)}
    x = torch.randn(2, 16, 64, generator=g)
    o = L.attention(w, "a", x, torch.zeros(2, 16, 64), 8)
    assert torch.allclose(o, w["a.to_out.0.bias"].expand_as(o), atol=1e-6)
