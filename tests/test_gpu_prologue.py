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


# ------------------------------------------------------------------------------- wav2vec2 audio encoder (§8f rank 2)
def test_wave_conv1d_and_time_axis_groupnorm_gelu():
    """vx_wave_conv1d (first feature-encoder conv on the raw fp32 waveform) and vx_groupnorm with one group per channel
    over the time axis + fused erf-GELU, against plain fp32 torch."""
    _need_gpu()
    import torch.nn.functional as F
    from v_express_amd import lib as L
    from v_express_amd import ops
    g = torch.Generator().manual_seed(0)
    for samples, c, taps, stride in ((4000, 32, 10, 5), (16001, 512, 10, 5), (57, 8, 16, 3), (10, 64, 10, 5)):
        wave = torch.randn(samples, generator=g)
        wt = torch.randn(taps, c, generator=g) * taps ** -0.5
        got = ops.wave_conv1d(wave.cuda(), wt.cuda(), stride)
        ref = wave.unfold(0, taps, stride) @ wt
        _close(got, ref, f"wave_conv1d {samples}x{c}", rel=4e-3, mx=2 ** -8)
        T = ref.shape[0]
        gamma, beta = torch.randn(c, generator=g) * 0.1 + 1, torch.randn(c, generator=g) * 0.1
        x = got.view(1, T, c)
        y = ops.groupnorm(x, gamma.cuda(), beta.cuda(), frames=1, hw=T, groups=c, eps=1e-5, silu=L.VX_ACT_GELU)
        if T > 1:
            yref = F.gelu(F.group_norm(x.float().cpu().transpose(1, 2), c, gamma, beta, 1e-5)).transpose(1, 2)
        else:                                     # one time step: zero variance, the output is GELU(beta)
            yref = F.gelu(beta).expand(1, 1, c)
        _close(y, yref, f"groupnorm(C groups)+gelu T={T}", rel=6e-3, mx=2 ** -7)
    with pytest.raises(L.VxError):
        ops.groupnorm(x, gamma.cuda(), beta.cuda(), frames=1, hw=T, groups=c, eps=1e-5, silu=3)


@pytest.mark.parametrize("T,c,k,s,n", [(799, 512, 3, 2, 512), (199, 512, 2, 2, 512), (799, 32, 3, 2, 32),
                                       (50, 64, 2, 2, 64), (3, 512, 3, 2, 512)])
def test_conv1d_as_gemm_over_overlapping_rows(T, c, k, s, n):
    """conv1d in the time-major layout == vx_gemm whose A rows overlap (row stride s*C < row length k*C): fast (K % 64
    == 0) and gather addressing paths, ragged M, GELU epilogue - against F.conv1d."""
    _need_gpu()
    import torch.nn.functional as F
    from v_express_amd import lib as L
    from v_express_amd import ops
    g = torch.Generator().manual_seed(T + c)
    h = torch.randn(T, c, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, c, k, generator=g) * (2.0 / (c * k)) ** 0.5).to(torch.bfloat16)
    t_out = (T - k) // s + 1
    hd = h.cuda()
    win = torch.as_strided(hd, (t_out, k * c), (s * c, 1))
    got = ops.gemm(win, w.permute(0, 2, 1).reshape(n, k * c).contiguous().cuda(), act=L.VX_ACT_GELU)
    ref = F.gelu(F.conv1d(h.float().t()[None], w.float(), stride=s))[0].t()
    _close(got, ref, f"conv1d-as-gemm T={T} c={c} k={k}", rel=6e-3, mx=2 ** -7)


@pytest.mark.parametrize("T,H,G,kp", [(18, 768, 16, 128), (249, 768, 16, 128), (12, 64, 4, 16)])
def test_grouped_positional_conv_as_per_group_gemms(T, H, G, kp):
    """Wav2Vec2PositionalConvEmbedding: x + GELU(conv1d(x, k, padding k/2, groups)[..., :-1]) through the group-major
    padded buffer + one overlapping-row GEMM per group writing a column slice with the residual fused."""
    _need_gpu()
    import torch.nn.functional as F
    from v_express_amd import synth
    from v_express_amd.wav2vec2 import Wav2Vec2Model
    g = torch.Generator().manual_seed(T)
    cfg = synth.Wav2Vec2Config(hidden_size=H, num_hidden_layers=0, num_attention_heads=4, conv_dim=(32,) * 7,
                               num_conv_pos_embeddings=kp, num_conv_pos_embedding_groups=G)
    sd = synth.wav2vec2_state_dict(cfg)
    m = Wav2Vec2Model(cfg).to("cuda")
    m.load_state_dict(sd)
    m._prepared()
    x = torch.randn(T, H, generator=g).to(torch.bfloat16)
    got = m._positional(x.cuda())
    from oracle import wav2vec2 as OW
    wp = OW.pos_conv_weight(sd).to(torch.bfloat16).float()
    pos = F.conv1d(x.float().t()[None], wp, sd["encoder.pos_conv_embed.conv.bias"], padding=kp // 2, groups=G)
    ref = x.float() + F.gelu(pos[0, :, :T]).t()
    _close(got, ref, f"positional conv T={T} H={H}", rel=6e-3, mx=2 ** -7)


@pytest.mark.parametrize("tag", ["small", "base"])
def test_wav2vec2_vs_oracle_and_transformers_golden(tag):
    _need_gpu()
    from oracle import wav2vec2 as OW
    from v_express_amd import Wav2Vec2Model, synth
    kw, samples = cases.W2V_CASES[tag]
    cfg = synth.Wav2Vec2Config(**kw)
    sd = synth.wav2vec2_state_dict(cfg)
    m = Wav2Vec2Model(cfg).to("cuda")
    m.load_state_dict(sd)
    wav = cases.waveform(samples)
    got = m(wav).last_hidden_state
    assert got.dtype == torch.float32 and got.shape == (1, cfg.num_frames(samples), cfg.hidden_size)
    ref = OW.forward(sd, wav, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.num_conv_pos_embedding_groups,
                     cfg.conv_stride, cfg.layer_norm_eps)
    _close(got, ref, f"Wav2Vec2Model[{tag}] vs oracle", rel=3e-2, mx=2 ** -4)
    gold = torch.load(os.path.join(GOLD, "wav2vec2.pt"), weights_only=False)[tag]
    _close(got, gold, f"Wav2Vec2Model[{tag}] vs transformers golden", rel=3e-2, mx=2 ** -4)
    feats = m.extract_features(wav[0].cuda())
    _close(feats, OW.feature_encoder(sd, wav, cfg.conv_stride)[0], f"feature encoder[{tag}]", rel=2e-2, mx=2 ** -4)
    if tag == "base":                                    # a second, longer clip: ragged T = 124 (2.5 s of audio)
        wav2 = cases.waveform(40000, seed=4)
        ref2 = OW.forward(sd, wav2, cfg.num_hidden_layers, cfg.num_attention_heads,
                          cfg.num_conv_pos_embedding_groups, cfg.conv_stride, cfg.layer_norm_eps)
        _close(m(wav2).last_hidden_state, ref2, "Wav2Vec2Model[base, 2.5 s] vs oracle", rel=3e-2, mx=2 ** -4)


def test_audio_path_waveform_to_audio_tokens_on_device():
    """prepare_audio_embeddings end to end on the HIP kernels: WaveformProcessor -> Wav2Vec2Model -> interpolation +
    windows -> AudioProjection, against the oracle chain (pipelines/v_express_pipeline.py:374-407)."""
    _need_gpu()
    import types
    import v_express_amd as vx
    from oracle import prologue as OP
    from oracle import wav2vec2 as OW
    from v_express_amd import synth
    wcfg = synth.Wav2Vec2Config(**cases.W2V_SMALL)
    wsd = synth.wav2vec2_state_dict(wcfg)
    enc = vx.Wav2Vec2Model(wcfg).to("cuda")
    enc.load_state_dict(wsd)
    acfg = synth.AudioProjectionConfig(dim=128, depth=2, dim_head=16, heads=8, num_queries=5,
                                       embedding_dim=wcfg.hidden_size, output_dim=128, max_seq_len=10)
    asd = synth.audio_projection_state_dict(acfg)
    proj = vx.AudioProjection(dim=acfg.dim, depth=acfg.depth, dim_head=acfg.dim_head, heads=acfg.heads,
                              num_queries=acfg.num_queries, embedding_dim=acfg.embedding_dim,
                              output_dim=acfg.output_dim, max_seq_len=acfg.max_seq_len).to("cuda")
    proj.load_state_dict(asd)
    raw = torch.randn(9600, generator=torch.Generator().manual_seed(8)) * 0.2 + 0.05         # 0.6 s @ 16 kHz
    F_, pad = 7, 2
    stub = types.SimpleNamespace(audio_processor=vx.WaveformProcessor(), audio_encoder=enc, audio_projection=proj,
                                 device=torch.device("cuda"))
    got = vx.VExpressPipeline.prepare_audio_embeddings(stub, raw, F_, pad, True)
    assert got.shape == (2, F_, 5, 128) and (got[0] == 0).all()
    wav = OW.normalize_waveform(raw)[None]
    states = OW.forward(wsd, wav, wcfg.num_hidden_layers, wcfg.num_attention_heads,
                        wcfg.num_conv_pos_embedding_groups, wcfg.conv_stride, wcfg.layer_norm_eps)
    ref = OP.audio_projection(asd, OP.audio_windows(states, F_, pad), acfg.depth, acfg.heads)
    _close(got[1], ref, "waveform -> audio tokens", rel=3e-2, mx=2 ** -4)


def test_pipeline_call_end_to_end_vs_oracle():
    """VExpressPipeline.__call__ from raw inputs (reference image, keypoint images, 16 kHz waveform) through every model
    on the HIP kernels - VAE encode, VKpsGuider, wav2vec2 + windows + AudioProjection, ReferenceNet banks, the windowed
    CFG loop, VAE decode - against the same chain built from the oracle pieces (tests/test_host_emulated.py holds the
    builders; the CPU suite runs the identical comparison over emulated kernels)."""
    _need_gpu()
    import test_host_emulated as E
    F_, steps, cf, co = 6, 2, 4, 2
    pipe, sd, cfgs = E.build_end_to_end("cuda")
    inp = E.end_to_end_inputs(F_)
    trace = []
    video = pipe(inp["ref_image"], inp["kps_images"], inp["waveform"], 64, 64, F_, steps, cases.GUIDANCE,
                 context_frames=cf, context_overlap=co, reference_attention_weight=cases.W_REF,
                 audio_attention_weight=cases.W_AUD, latents=inp["latents"],
                 callback=lambda i, t, l: trace.append(l.detach().cpu().clone()))
    with torch.no_grad():
        lat_ref, video_ref = E.oracle_end_to_end(sd, cfgs, inp, F_, steps, cf, co)
    assert video.device.type == "cpu" and video.shape == video_ref.shape
    err = (trace[-1] - lat_ref).norm() / lat_ref.norm()
    mae = (video - video_ref).abs().mean().item()
    print(f"[end to end] latents relL2={err:.4g} video MAE={mae:.4g}")
    assert err <= 5e-2 and mae <= 2e-2, (err, mae)
