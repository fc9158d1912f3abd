"""Once-per-clip prologue + post-processing on a real MI355X: VKpsGuider,
AudioProjection, VAE encode (HIP kernels through the C ABI) against the fp32 CPU oracle (oracle/prologue.py, pinned to
the reference by tests/golden/prologue.pt) and against the reference's golden outputs directly; the median filter and
the uint8 frame packing are selections / exact fp32 products, so they must be BIT-EXACT.

Tolerances (bf16 storage, fp32 accumulate vs fp32): relative L2 <= 2e-2, max |err| <= 2^-5 * max |ref| through the
7-conv guider / 4-block projector / 20-layer encoder (the per-kernel bound of test_gpu_kernels.py compounded).
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


def _close(got, ref, what, rel=2e-2, mx=2 ** -5):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    err = (got - ref).abs()
    rl2 = (err.pow(2).sum().sqrt() / (ref.pow(2).sum().sqrt() + 1e-12)).item()
    assert rl2 <= rel and err.max().item() <= mx * ref.abs().max().item() + 1e-5, \
        f"{what}: relL2={rl2:.4g} max|err|={err.max().item():.4g} max|ref|={ref.abs().max().item():.4g}"


def _gold():
    return torch.load(os.path.join(GOLD, "prologue.pt"), weights_only=False)


@pytest.mark.parametrize("tag,kw", [("small", cases.KPS_SMALL), ("full", {})])
def test_kps_guider_vs_oracle_and_reference_golden(tag, kw):
    _need_gpu()
    from oracle import prologue as OP
    from v_express_amd import VKpsGuider, synth
    kcfg = synth.KpsGuiderConfig(**kw)
    sd = synth.kps_guider_state_dict(kcfg)
    inp = cases.prologue_inputs()
    m = VKpsGuider(kcfg.conditioning_embedding_channels, block_out_channels=kcfg.block_out_channels).to("cuda")
    m.load_state_dict(sd)
    got = m(inp["kps_images"])
    b, c, f, H, W = inp["kps_images"].shape
    ref = OP.kps_guider(sd, inp["kps_images"].permute(0, 2, 1, 3, 4).reshape(b * f, c, H, W))
    ref = ref.reshape(b, f, -1, H // 8, W // 8).permute(0, 2, 1, 3, 4)
    _close(got, ref, f"VKpsGuider[{tag}] vs oracle")
    _close(got, _gold()[f"kps_{tag}"], f"VKpsGuider[{tag}] vs reference golden")
    tok, h, w = m.forward_tokens(inp["kps_images"])
    assert (h, w) == (H // 8, W // 8) and tok.shape == (b * f, h * w, kcfg.conditioning_embedding_channels)
    assert torch.equal(tok.float().view(b, f, h, w, -1).permute(0, 4, 1, 2, 3), got)


@pytest.mark.parametrize("tag,kw,key", [("small", cases.AUDIO_SMALL, "audio_windows_small"),
                                        ("full", {}, "audio_windows_full")])
def test_audio_projection_vs_oracle_and_reference_golden(tag, kw, key):
    _need_gpu()
    from oracle import prologue as OP
    from v_express_amd import AudioProjection, synth
    acfg = synth.AudioProjectionConfig(**kw)
    sd = synth.audio_projection_state_dict(acfg)
    inp = cases.prologue_inputs()
    m = AudioProjection(dim=acfg.dim, depth=acfg.depth, dim_head=acfg.dim_head, heads=acfg.heads,
                        num_queries=acfg.num_queries, embedding_dim=acfg.embedding_dim, output_dim=acfg.output_dim,
                        ff_mult=acfg.ff_mult, max_seq_len=acfg.max_seq_len).to("cuda")
    m.load_state_dict(sd)
    got = m(inp[key])
    _close(got, OP.audio_projection(sd, inp[key], acfg.depth, acfg.heads), f"AudioProjection[{tag}] vs oracle")
    _close(got, _gold()[f"audio_{tag}"], f"AudioProjection[{tag}] vs reference golden")


def test_vae_encode_vs_oracle_and_reference_golden():
    _need_gpu()
    import oracle
    from oracle import prologue as OP
    from v_express_amd import AutoencoderKL, synth
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    sd = synth.vae_encoder_state_dict(vcfg)
    inp = cases.prologue_inputs()
    vae = AutoencoderKL(vcfg).to("cuda")
    vae.load_state_dict(sd)
    got = vae.encode(inp["ref_image"]).latent_dist.mean
    ref = OP.vae_encode_mean(sd, oracle.VaeConfig(**cases.SMALL_VAE), inp["ref_image"])
    _close(got, ref, "VAE encode mean vs oracle", rel=3e-2)
    _close(got, _gold()["vae_mean"], "VAE encode mean vs reference golden", rel=3e-2)


@pytest.mark.parametrize("shape", [(3, 5, 12, 10), (3, 2, 2, 2), (1, 9, 33, 65), (3, 16, 64, 64)])
def test_median_filter_and_uint8_pack_bit_exact(shape):
    _need_gpu()
    from oracle import prologue as OP
    from v_express_amd import median_filter_3d, video_frames_uint8
    if shape == (3, 5, 12, 10):
        video = cases.prologue_inputs()["video"]
    else:
        video = torch.rand(shape, generator=torch.Generator().manual_seed(sum(shape)))
        video[:, :, ::2] = (video[:, :, ::2] * 8).round() / 8                                    # many ties
    want = OP.median_filter_3d(video, 3)
    got = median_filter_3d(video, 3, "cuda")
    assert got.device == video.device and torch.equal(got, want)
    u8 = video_frames_uint8(video.cuda()[None])
    assert torch.equal(u8.cpu(), torch.from_numpy(OP.frames_uint8(want)))
    if shape == (3, 5, 12, 10):
        g = _gold()
        assert torch.equal(got, g["median"]) and torch.equal(u8.cpu(), g["median_u8"])


def test_pipeline_call_runs_prologue_on_device():
    """VExpressPipeline.__call__ with image / waveform inputs: the prologue hooks (VAE encode, VKpsGuider tokens,
    audio windows + AudioProjection) produce exactly what the same modules give when called by hand and passed in
    through the keyword arguments (bit-identical latents)."""
    _need_gpu()
    import types
    import v_express_amd as vx
    from v_express_amd import synth
    from v_express_amd.prologue import audio_windows
    cfg = cases.unet_cfg(cases.SMALL)
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    unet = vx.UNet3DConditionModel(cfg).to("cuda")
    refnet = vx.UNet2DConditionModel(cfg).to("cuda")
    unet.load_state_dict(synth.unet3d_state_dict(cfg))
    refnet.load_state_dict(synth.refnet_state_dict(cfg))
    vae = vx.AutoencoderKL(vcfg).to("cuda")
    vae.load_state_dict({**synth.vae_decoder_state_dict(vcfg), **synth.vae_encoder_state_dict(vcfg)})
    kcfg = synth.KpsGuiderConfig(**cases.KPS_SMALL)
    guider = vx.VKpsGuider(kcfg.conditioning_embedding_channels, block_out_channels=kcfg.block_out_channels).to("cuda")
    guider.load_state_dict(synth.kps_guider_state_dict(kcfg))
    acfg = synth.AudioProjectionConfig(dim=128, depth=2, dim_head=16, heads=8, num_queries=5, embedding_dim=96,
                                       output_dim=768, max_seq_len=10)
    proj = vx.AudioProjection(dim=acfg.dim, depth=acfg.depth, dim_head=acfg.dim_head, heads=acfg.heads,
                              num_queries=acfg.num_queries, embedding_dim=acfg.embedding_dim, output_dim=acfg.output_dim,
                              max_seq_len=acfg.max_seq_len).to("cuda")
    proj.load_state_dict(synth.audio_projection_state_dict(acfg))
    sched = vx.DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                             steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True,
                             timestep_spacing="trailing")
    g = torch.Generator().manual_seed(3)
    F_, H, W = 6, 64, 64
    states = torch.randn(1, 29, 96, generator=g)

    class Enc:
        def parameters(self):
            return iter([torch.zeros(1, dtype=torch.float32)])

        def __call__(self, wav):
            return types.SimpleNamespace(last_hidden_state=states.to(wav.device))
    pipe = vx.VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet, v_kps_guider=guider,
                               audio_processor=lambda wav, return_tensors, sampling_rate: {"input_values": wav},
                               audio_encoder=Enc(), audio_projection=proj, scheduler=sched)
    ref_img = torch.rand(1, 3, H, W, generator=g)
    kps = [torch.rand(1, 3, H, W, generator=g) for _ in range(F_)]
    lat0 = torch.randn(1, 4, F_, H // 8, W // 8, generator=g)
    kw = dict(width=W, height=H, video_length=F_, num_inference_steps=2, guidance_scale=3.5, context_frames=4,
              context_overlap=2, reference_attention_weight=0.95, audio_attention_weight=3.0, latents=lat0,
              decode=False)
    a = pipe(ref_img, kps, torch.zeros(1, 16), **kw)
    # by hand
    ref_lat = vae.encode(2.0 * ref_img - 1.0).latent_dist.mean * 0.18215
    feat = guider(torch.cat([k.unsqueeze(2) for k in kps], dim=2))
    feat = torch.cat([torch.zeros_like(feat), feat], dim=0)
    aud = proj(audio_windows(states.cuda(), F_, 2)).unsqueeze(0)
    aud = torch.cat([torch.zeros_like(aud), aud], dim=0)
    b = pipe(None, None, None, reference_latents=ref_lat, kps_features=feat, audio_embeddings=aud, **kw)
    assert torch.isfinite(a).all() and torch.equal(a, b)
