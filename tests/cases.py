"""Shared seeded test cases (TEST INFRASTRUCTURE): configs + synthetic tensors used by the golden
generator (dev container, Tier A), the oracle tests (CPU, anywhere) and the GPU parity tests."""
from v_express_amd import synth

SMALL = dict(block_out_channels=(64, 128, 256, 256))          # head dims 8/16/32, 57.7 M params
FULL = dict(block_out_channels=(320, 640, 1280, 1280))        # the SD-1.5 widths (head dims 40/80/160)
SMALL_VAE = dict(block_out_channels=(32, 64, 128, 128))

W_REF, W_AUD, GUIDANCE = 0.95, 3.0, 3.5                        # inference.py:66,69-70 defaults

# name -> (unet cfg kwargs, F, latent h, latent w, timestep)
FORWARD_CASES = {
    "small_f4_8x8": (SMALL, 4, 8, 8, 959),
    "small_f8_16x8": (SMALL, 8, 16, 8, 39),
    "full_f4_8x8": (FULL, 4, 8, 8, 959),
}

# name -> (F, context_frames, context_overlap, steps)   (SMALL unet + SMALL_VAE, 8x8 latents)
PIPELINE_CASES = {
    "aligned_F10_c4o2": (10, 4, 2, 3),
    "reflected_F11_c4o2": (11, 4, 2, 3),       # last window [8,9,10,9]: SURVEY.md Appendix D #10
    "single_F8_c8o2": (8, 8, 2, 3),
}


# guidance_scale <= 1: no classifier-free guidance - batch of 1, conditional inputs only, bank without the zero half
NOCFG_CASE = ("nocfg_F10_c4o2", 10, 4, 2, 3)          # name, F, context_frames, context_overlap, steps


# BASELINE.json configs[1] through the reference itself (make_golden.py fullsize): F, ctx frames, overlap, steps
FULLSIZE_CASE = (16, 16, 4, 25)
FULLSIZE_FRAMES = (0, 15)                             # decoded 512x512 frames kept in the golden file
# the sliding-window path at the benchmarked geometry (make_golden.py fullsize_F28): two overlapping 16-frame windows
FULLSIZE_F28_CASE = (28, 16, 4, 2)
# the reference's DEFAULT window (inference.py:67-68: context_frames 24, context_overlap 4) at the benchmarked geometry
# (make_golden.py fullsize_ctx24): F = 44 = windows [0..23] and [20..43] sharing four frames, 2 DDIM steps
FULLSIZE_CTX24_CASE = (44, 24, 4, 2)
# BASELINE configs[4]'s geometry (768x768 = 96x96 latents) through the reference (make_golden.py fullsize_768)
FULLSIZE_768_CASE = (4, 4, 2, 2)


def cond_only(inp):
    """The conditional half of synthetic_inputs' CFG pairs (what the prologue hooks return without CFG)."""
    return dict(inp, kps_features=inp["kps_features"][1:], audio_embeddings=inp["audio_embeddings"][1:])


def unet_cfg(kw):
    return synth.UNetConfig(**kw)


def oracle_cfg(kw):
    import oracle
    return oracle.UNetConfig(**kw)


# ---- once-per-clip prologue / post-processing: seeded inputs shared by make_golden.py,
# the oracle tests and the GPU parity tests
KPS_SMALL = dict(conditioning_embedding_channels=64, block_out_channels=(16, 32, 48, 64))
AUDIO_SMALL = dict(dim=128, depth=2, dim_head=16, heads=8, num_queries=5, embedding_dim=96, output_dim=128,
                   max_seq_len=10)


def prologue_inputs(seed=11):
    import torch
    g = torch.Generator().manual_seed(seed)
    return dict(
        kps_images=torch.rand(1, 3, 3, 64, 48, generator=g),            # [b, 3, f, H, W] in [0, 1]
        audio_windows_small=torch.randn(4, 10, 96, generator=g),       # [F, 2*(2*pad+1), d]
        audio_windows_full=torch.randn(3, 10, 768, generator=g),
        wav2vec_states=torch.randn(1, 37, 96, generator=g),            # [1, T, d] "last_hidden_state"
        ref_image=torch.rand(1, 3, 64, 48, generator=g) * 2 - 1,        # [-1, 1]
        video=torch.rand(3, 5, 12, 10, generator=g),                    # [C, F, H, W] in [0, 1]
    )


# ---- wav2vec2 audio encoder: configs + seeded waveform shared by make_golden.py (transformers'
# own Wav2Vec2Model), the oracle tests and the GPU parity tests
W2V_SMALL = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, conv_dim=(32,) * 7,
                 num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
W2V_CASES = {"small": (W2V_SMALL, 4000), "base": ({}, 6000)}          # name -> (config kwargs, samples @ 16 kHz)


def waveform(samples, seed=3):
    """Seeded raw audio, already through the processor's zero-mean / unit-variance normalisation: [1, samples]."""
    import torch
    wav = torch.randn(1, samples, generator=torch.Generator().manual_seed(seed)) * 0.1
    return (wav - wav.mean()) / torch.sqrt(wav.var(unbiased=False) + 1e-7)


def hf_wav2vec2(cfg, sd):
    """transformers' own Wav2Vec2Model with the synthetic weights loaded strictly (pins the key schema)."""
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    hf = Wav2Vec2Model(Wav2Vec2Config(**{k: (list(v) if isinstance(v, tuple) else v)
                                         for k, v in cfg.__dict__.items()})).eval()
    hf.load_state_dict(sd, strict=True)
    return hf
