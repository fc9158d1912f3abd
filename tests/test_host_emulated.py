"""The whole HOST side on CPU: models, denoising loop and multi-rank sharding run with tests/fake_ops.py standing in for
the HIP wrappers (fp32 torch arithmetic, bf16 rounding at every kernel boundary like the real kernels) and are compared
with the reference's golden outputs and the fp32 oracle.  This checks everything that is not a kernel - weight
re-layouts (NHWC conv matrices, fused QKV, GEGLU interleave), the padded-image convolutions, bank handling and the CFG
shortcuts, window / overlap bookkeeping, unit and frame sharding with their collectives - in the dev container, where
no GPU exists.  The kernels themselves, and the same host code on top of them, are what the `-m gpu` tests check."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import cases
import dist_gpu_worker as W

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.fixture()
def emulated(monkeypatch):
    import fake_ops
    from v_express_amd import ops, prologue, unet_3d, vae
    fake_ops.install(monkeypatch, ops)
    monkeypatch.setattr(unet_3d._UNetBase, "_need_gpu", lambda self: None)
    monkeypatch.setattr(vae.AutoencoderKLDecoder, "_need_gpu", lambda self: None)
    monkeypatch.setattr(prologue._Module, "_need_gpu", lambda self: None)
    monkeypatch.setattr(ops, "_PADDED", {})
    return ops


@pytest.mark.parametrize("name", ["small_f4_8x8", "small_f8_16x8", "full_f4_8x8"])      # full = the SD-1.5 widths
def test_unet_forward_host_composition_vs_reference_golden(emulated, name):
    import v_express_amd as vx
    from oracle import unet as OU
    from v_express_amd import synth
    kw, F, h, w, t = cases.FORWARD_CASES[name]
    cfg, ocfg = cases.unet_cfg(kw), cases.oracle_cfg(kw)
    sd3, sd2 = synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg)
    inp = synth.synthetic_inputs(cfg, F, h, w)
    unet, refnet = vx.UNet3DConditionModel(cfg).to("cpu"), vx.UNet2DConditionModel(cfg).to("cpu")
    unet.load_state_dict(sd3, strict=True)
    refnet.load_state_dict(sd2, strict=True)
    writer = vx.ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    reader = vx.ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                          reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
    obanks = OU.refnet_banks(sd2, ocfg, inp["ref_latents"])
    assert sorted(refnet.banks) == sorted(obanks)
    assert max(rel_l2(refnet.banks[k].view_as(obanks[k][0]), obanks[k][0]) for k in obanks) <= 3e-2
    reader.update(writer, True)
    x = inp["latents"].repeat(2, 1, 1, 1, 1)
    ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
    got = unet(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
    gold = torch.load(os.path.join(GOLD, f"forward_{name}.pt"), weights_only=False)["pred"]
    assert got.shape == gold.shape and rel_l2(got, gold) <= 3e-2
    # BASELINE.json configs[4]: the attention q / k / v / out projections on per-row e4m3 operands.  Stated tolerance of
    # the fp8 mode against the fp32 reference: one CFG forward rel-L2 <= 6e-2, cosine >= 0.998 (e4m3 carries 3 mantissa
    # bits: ~2^-4 relative per element, averaged down by the K-long dot products; measured here ~2e-2).
    unet.fp8_projections = True
    got8 = unet(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
    r8 = rel_l2(got8, gold)
    c8 = torch.nn.functional.cosine_similarity(got8.flatten().double(), gold.flatten().double(), dim=0).item()
    print(f"[{name}] fp8 projections: relL2={r8:.4g} cosine={c8:.6f} (bf16: {rel_l2(got, gold):.4g})")
    assert not torch.equal(got8, got) and r8 <= 6e-2 and c8 >= 0.998, (r8, c8)


def test_round3_fused_paths_host_composition_vs_reference_golden(emulated, monkeypatch):
    """The host code of the round-3 fusions, forced on at a size where the routing rules would not pick them
    (`ops.gn_fold_applies`, `ops.qk_on_ring` want the 64x64-level launch sizes): GroupNorm folded into proj_in through
    per-frame weights, Q | K as one GEMM with column views + V^T alone, LayerNorm statistics taken from the producing
    GEMM (`stats_out`), the unconditional half's constant terms as a per-item row bias.  Same reference golden, same
    bound as the default composition, and both compositions within bf16 rounding of each other."""
    import v_express_amd as vx
    from v_express_amd import ops, synth
    name = "small_f4_8x8"
    kw, F, h, w, t = cases.FORWARD_CASES[name]
    cfg = cases.unet_cfg(kw)
    unet, refnet = vx.UNet3DConditionModel(cfg).to("cpu"), vx.UNet2DConditionModel(cfg).to("cpu")
    unet.load_state_dict(synth.unet3d_state_dict(cfg), strict=True)
    refnet.load_state_dict(synth.refnet_state_dict(cfg), strict=True)
    inp = synth.synthetic_inputs(cfg, F, h, w)
    writer = vx.ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    reader = vx.ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                          reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
    reader.update(writer, True)
    x = inp["latents"].repeat(2, 1, 1, 1, 1)
    ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
    gold = torch.load(os.path.join(GOLD, f"forward_{name}.pt"), weights_only=False)["pred"]
    base = unet(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
    used = {"fold": 0, "qk": 0}

    def fold_everywhere(m, hw, c, n):
        used["fold"] += 1
        return True

    def qk_everywhere(m, c):
        used["qk"] += 1
        return True
    def tblock_everywhere(c, heads, f, hw):
        used["tblock"] = used.get("tblock", 0) + 1
        return True
    monkeypatch.setattr(ops, "gn_fold_applies", fold_everywhere)
    monkeypatch.setattr(ops, "qk_on_ring", qk_everywhere)
    monkeypatch.setattr(ops, "tblock_fused_applies", tblock_everywhere)
    fused = unet(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
    assert used["fold"] == 16 + 21 and used["qk"] == 16          # every spatial + motion proj_in, every spatial qkv
    assert used["tblock"] == 2 * 21                              # both attention blocks of every motion module (round 4)
    print(f"[round-3 fused composition] vs golden {rel_l2(fused, gold):.4g} (default {rel_l2(base, gold):.4g}), "
          f"fused vs default {rel_l2(fused, base):.4g}")
    assert rel_l2(fused, gold) <= 3e-2 and rel_l2(base, gold) <= 3e-2
    assert not torch.equal(fused, base) and rel_l2(fused, base) <= 2e-2
    # round 4: row statistics as two-part sums ([rows, 4] buffers; vx_gemm_params.row_stats_parts / ln_stats_parts), forced
    # on at every width of this model - producers write the halves' sums, consumers finish them
    made = []
    real_buffer = ops.stats_buffer

    def counting_buffer(rows, c, device):
        made.append(real_buffer(rows, c, device).shape[1])
        return real_buffer(rows, c, device)
    monkeypatch.setattr(ops, "STATS_PARTS_WIDTHS", set(range(8, 4096, 8)))
    monkeypatch.setattr(ops, "stats_buffer", counting_buffer)
    parts = unet(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0]
    assert made and all(w_ == 4 for w_ in made)
    print(f"[two-part row statistics] vs one-part {rel_l2(parts, fused):.4g}, vs golden {rel_l2(parts, gold):.4g}")
    assert rel_l2(parts, fused) <= 2e-2 and rel_l2(parts, gold) <= 3e-2      # (bf16 rounding flips, like fused vs default)


@pytest.mark.parametrize("name", ["aligned_F10_c4o2", "reflected_F11_c4o2", cases.NOCFG_CASE[0]])
def test_denoising_loop_and_decode_host_composition_vs_reference_golden(emulated, name):
    import v_express_amd as vx
    from v_express_amd import ops, synth
    from v_express_amd.context import get_context_scheduler
    import ref_import as R
    do_cfg = name != cases.NOCFG_CASE[0]            # the last case: guidance_scale = 1.0, batch of 1
    Fn, cf, co, steps = cases.PIPELINE_CASES[name] if do_cfg else cases.NOCFG_CASE[1:]
    guidance = cases.GUIDANCE if do_cfg else 1.0
    cfg = cases.unet_cfg(cases.SMALL)
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    unet, refnet = vx.UNet3DConditionModel(cfg).to("cpu"), vx.UNet2DConditionModel(cfg).to("cpu")
    unet.load_state_dict(synth.unet3d_state_dict(cfg), strict=True)
    refnet.load_state_dict(synth.refnet_state_dict(cfg), strict=True)
    vae = vx.AutoencoderKLDecoder(vcfg).to("cpu")
    vae.load_state_dict(synth.vae_decoder_state_dict(vcfg))
    sched = vx.DDIMScheduler(**R.NOISE_SCHEDULER_KWARGS)
    pipe = vx.VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet, scheduler=sched)
    inp = synth.synthetic_inputs(cfg, Fn, 8, 8)
    if not do_cfg:
        inp = cases.cond_only(inp)
    nb = inp["kps_features"].shape[0]
    writer = vx.ReferenceAttentionControl(refnet, do_classifier_free_guidance=do_cfg, mode="write",
                                          fusion_blocks="full")
    reader = vx.ReferenceAttentionControl(unet, do_classifier_free_guidance=do_cfg, mode="read", fusion_blocks="full",
                                          reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
    reader.update(writer, do_cfg)
    sched.set_timesteps(steps)
    windows = list(get_context_scheduler("uniform")(step=0, num_frames=Fn, context_size=cf, context_stride=1,
                                                    context_overlap=co, closed_loop=False))
    c0 = cfg.block_out_channels[0]
    kps = ops.ncfhw_to_nhwc(inp["kps_features"], c0).view(nb, Fn, 64, c0)
    audio = inp["audio_embeddings"].to(torch.bfloat16).contiguous()
    lat = inp["latents"].clone().float()
    if not do_cfg:
        with pytest.raises(ValueError):                        # CFG-shaped inputs with guidance_scale <= 1
            pipe.denoise(lat.clone(), torch.cat([kps, kps]), torch.cat([audio, audio]), sched.timesteps.tolist(),
                         windows, guidance)
    pipe.denoise(lat, kps, audio, sched.timesteps.tolist(), windows, guidance)
    g = torch.load(os.path.join(GOLD, f"pipeline_{name}.pt"), weights_only=False)
    assert rel_l2(lat, g["latents"]) <= 5e-2
    video = pipe.decode_latents(lat)
    assert video.shape == g["video_f16"].shape
    assert (video - g["video_f16"].float()).abs().mean().item() <= 2e-2


def test_prologue_models_host_composition_vs_reference_golden(emulated):
    import v_express_amd as vx
    from v_express_amd import synth
    g = torch.load(os.path.join(GOLD, "prologue.pt"), weights_only=False)
    inp = cases.prologue_inputs()
    kcfg = synth.KpsGuiderConfig(**cases.KPS_SMALL)
    m = vx.VKpsGuider(kcfg.conditioning_embedding_channels, block_out_channels=kcfg.block_out_channels).to("cpu")
    m.load_state_dict(synth.kps_guider_state_dict(kcfg))
    assert rel_l2(m(inp["kps_images"]), g["kps_small"]) <= 2e-2
    acfg = synth.AudioProjectionConfig(**cases.AUDIO_SMALL)
    a = vx.AudioProjection(dim=acfg.dim, depth=acfg.depth, dim_head=acfg.dim_head, heads=acfg.heads,
                           num_queries=acfg.num_queries, embedding_dim=acfg.embedding_dim, output_dim=acfg.output_dim,
                           max_seq_len=acfg.max_seq_len).to("cpu")
    a.load_state_dict(synth.audio_projection_state_dict(acfg))
    assert rel_l2(a(inp["audio_windows_small"]), g["audio_small"]) <= 2e-2
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    v = vx.AutoencoderKL(vcfg).to("cpu")
    v.load_state_dict({**synth.vae_decoder_state_dict(vcfg), **synth.vae_encoder_state_dict(vcfg)})
    assert rel_l2(v.encode(inp["ref_image"]).latent_dist.mean, g["vae_mean"]) <= 2e-2


def _worker(rank, world, port, F, cf, co, S, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W.emulate_kernels()
    lat = W.run(F, cf, co, 2, S, device="cpu")
    q.put((rank, lat.float().numpy().copy()))      # by value: a tensor's file descriptor must be fetched while this process lives
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,S,F,cf,co", [(2, 2, 8, 8, 2), (4, 0, 8, 8, 2), (2, 1, 14, 8, 2), (4, 1, 32, 8, 2)])
def test_sharded_loop_with_the_real_host_code_matches_single_process(emulated, world, S, F, cf, co):
    """`VExpressPipeline.denoise` + `UNet3DConditionModel.forward_tokens` + `blocks.motion_module(shard=)` under real
    process groups (gloo): unit sharding (S = 1), frame sharding (S = 2) and the automatic policy (S = 0: 4 ranks, one
    window -> 2 ranks per unit) against the single-process loop.  Like the real kernels, the emulated ones (float64
    inside, bf16 at the boundaries) give a row the same result whatever else shares the call, so sharding is pure data
    movement and the clip must come out BIT-identical, on every rank."""
    ref = W.run(F, cf, co, 2, 0, device="cpu")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, F, cf, co, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [(r, torch.from_numpy(a)) for r, a in (q.get(timeout=600) for _ in procs)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, lat in results:
        assert torch.isfinite(lat).all() and torch.equal(lat, ref), (rank, rel_l2(lat, ref))


@pytest.mark.parametrize("world,S,F,cf,co", [(2, 2, 8, 8, 2), (2, 1, 14, 8, 2)])
def test_sharded_loop_with_the_round4_paths_forced_on(emulated, monkeypatch, world, S, F, cf, co):
    """The same comparison with the round-4 host paths forced on in every process (`dist_gpu_worker.force_round4_paths`):
    the temporal attention blocks as single `ops.tblock_fused` calls - in the frame-sharded case on the pixel-shard layout
    between the two all-to-alls - and two-part row statistics.  Sharding stays pure data movement: BIT-identical clips, and
    the forced paths really ran (the clip differs from the default composition's by rounding only)."""
    from v_express_amd import ops
    default = W.run(F, cf, co, 2, 0, device="cpu")
    W.force_round4_paths(ops, monkeypatch.setattr)
    monkeypatch.setenv("VX_TEST_FORCE_ROUND4", "1")
    ref = W.run(F, cf, co, 2, 0, device="cpu")
    assert not torch.equal(ref, default) and rel_l2(ref, default) <= 3e-2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, F, cf, co, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [(r, torch.from_numpy(a)) for r, a in (q.get(timeout=600) for _ in procs)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, lat in results:
        assert torch.isfinite(lat).all() and torch.equal(lat, ref), (rank, rel_l2(lat, ref))


def build_end_to_end(device):
    """Every model of VExpressPipeline.__call__ at its small test configuration with seeded synthetic weights:
    (pipeline, state dicts, configs)."""
    import v_express_amd as vx
    from v_express_amd import synth
    import ref_import as R
    cfg = cases.unet_cfg(cases.SMALL)
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    kcfg = synth.KpsGuiderConfig(**cases.KPS_SMALL)
    wcfg = synth.Wav2Vec2Config(**cases.W2V_SMALL)
    acfg = synth.AudioProjectionConfig(dim=128, depth=2, dim_head=16, heads=8, num_queries=5,
                                       embedding_dim=wcfg.hidden_size, output_dim=cfg.cross_attention_dim, max_seq_len=10)
    sd = dict(unet=synth.unet3d_state_dict(cfg), refnet=synth.refnet_state_dict(cfg),
              vae={**synth.vae_decoder_state_dict(vcfg), **synth.vae_encoder_state_dict(vcfg)},
              kps=synth.kps_guider_state_dict(kcfg), w2v=synth.wav2vec2_state_dict(wcfg),
              proj=synth.audio_projection_state_dict(acfg))
    unet, refnet = vx.UNet3DConditionModel(cfg).to(device), vx.UNet2DConditionModel(cfg).to(device)
    unet.load_state_dict(sd["unet"], strict=True)
    refnet.load_state_dict(sd["refnet"], strict=True)
    vae = vx.AutoencoderKL(vcfg).to(device)
    vae.load_state_dict(sd["vae"])
    guider = vx.VKpsGuider(kcfg.conditioning_embedding_channels, block_out_channels=kcfg.block_out_channels).to(device)
    guider.load_state_dict(sd["kps"])
    enc = vx.Wav2Vec2Model(wcfg).to(device)
    enc.load_state_dict(sd["w2v"])
    proj = vx.AudioProjection(dim=acfg.dim, depth=acfg.depth, dim_head=acfg.dim_head, heads=acfg.heads,
                              num_queries=acfg.num_queries, embedding_dim=acfg.embedding_dim,
                              output_dim=acfg.output_dim, max_seq_len=acfg.max_seq_len).to(device)
    proj.load_state_dict(sd["proj"])
    pipe = vx.VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet, v_kps_guider=guider,
                               audio_processor=vx.WaveformProcessor(), audio_encoder=enc, audio_projection=proj,
                               scheduler=vx.DDIMScheduler(**R.NOISE_SCHEDULER_KWARGS))
    return pipe, sd, dict(unet=cfg, vae=vcfg, w2v=wcfg, proj=acfg)


def end_to_end_inputs(F_=6, size=64, seed=21):
    g = torch.Generator().manual_seed(seed)
    return dict(ref_image=torch.rand(1, 3, size, size, generator=g),
                kps_images=[torch.rand(1, 3, size, size, generator=g) for _ in range(F_)],
                waveform=torch.randn(9600, generator=g) * 0.2 + 0.05,                   # 0.6 s @ 16 kHz
                latents=torch.randn(1, 4, F_, size // 8, size // 8, generator=g))


def oracle_end_to_end(sd, cfgs, inp, F_, steps, cf, co, pad=2):
    """The reference's __call__ (pipelines/v_express_pipeline.py:409-589) restated with the oracle pieces."""
    import oracle
    from oracle import loop as OL, prologue as OP, unet as OU, vae as OV, wav2vec2 as OW
    ocfg = cases.oracle_cfg(cases.SMALL)
    ovcfg = oracle.VaeConfig(**cases.SMALL_VAE)
    ref_lat = OP.vae_encode_mean(sd["vae"], ovcfg, 2.0 * inp["ref_image"] - 1.0) * 0.18215           # :343-348
    kps = OP.kps_guider(sd["kps"], torch.cat(inp["kps_images"], dim=0))                                # [F, C, h, w]
    kps = kps.permute(1, 0, 2, 3)[None]
    kps = torch.cat([torch.zeros_like(kps), kps], dim=0)                                               # :368-371
    wc = cfgs["w2v"]
    states = OW.forward(sd["w2v"], OW.normalize_waveform(inp["waveform"])[None], wc.num_hidden_layers,
                        wc.num_attention_heads, wc.num_conv_pos_embedding_groups, wc.conv_stride, wc.layer_norm_eps)
    aud = OP.audio_projection(sd["proj"], OP.audio_windows(states, F_, pad), cfgs["proj"].depth,
                              cfgs["proj"].heads)[None]
    aud = torch.cat([torch.zeros_like(aud), aud], dim=0)                                               # :403-405
    banks = OU.reader_banks(OU.refnet_banks(sd["refnet"], ocfg, ref_lat))
    ddim = OL.DDIM()
    lat = OL.mean_overlap(lambda x, t, e, k: OU.unet3d_forward(sd["unet"], ocfg, x, t, e, k, banks, cases.W_REF,
                                                               cases.W_AUD),
                          inp["latents"], ddim.set_timesteps(steps), ddim, OL.uniform_windows(F_, cf, co),
                          cases.GUIDANCE, kps, aud)
    return lat, OV.decode_latents(sd["vae"], ovcfg, lat)


def test_pipeline_call_end_to_end_host_composition_vs_oracle(emulated):
    """VExpressPipeline.__call__ from raw inputs - reference image, keypoint images, 16 kHz waveform - through every
    model (VAE encode, VKpsGuider, wav2vec2 + windows + AudioProjection, ReferenceNet banks, the windowed CFG loop, VAE
    decode) against the same chain built from the oracle pieces."""
    F_, steps, cf, co = 6, 2, 4, 2
    pipe, sd, cfgs = build_end_to_end("cpu")
    inp = end_to_end_inputs(F_)
    trace = []
    video = pipe(inp["ref_image"], inp["kps_images"], inp["waveform"], 64, 64, F_, steps, cases.GUIDANCE,
                 context_frames=cf, context_overlap=co, reference_attention_weight=cases.W_REF,
                 audio_attention_weight=cases.W_AUD, latents=inp["latents"], output_device=None,
                 callback=lambda i, t, l: trace.append(l.detach().clone()))
    with torch.no_grad():
        lat_ref, video_ref = oracle_end_to_end(sd, cfgs, inp, F_, steps, cf, co)
    assert video.shape == video_ref.shape == (1, 3, F_, 64, 64)
    assert rel_l2(trace[-1], lat_ref) <= 5e-2
    assert (video - video_ref).abs().mean().item() <= 2e-2


def test_merging_windows_into_one_unet_call_changes_nothing(emulated):
    """`VExpressPipeline.units_per_call`: one window per UNet call (2) vs up to three (6); the default is 4.  Batch rows are
    independent in every kernel, so the clip must be bit-identical to the default one-window-per-call loop."""
    import v_express_amd as vx
    orig = vx.VExpressPipeline.__init__

    def unmerged(self, *a, **k):
        orig(self, *a, **k)
        self.units_per_call = 2
    vx.VExpressPipeline.__init__ = unmerged
    try:
        ref = W.run(14, 8, 2, 2, 0, device="cpu")
    finally:
        vx.VExpressPipeline.__init__ = orig

    def patched(self, *a, **k):
        orig(self, *a, **k)
        self.units_per_call = 6
    vx.VExpressPipeline.__init__ = patched
    try:
        got = W.run(14, 8, 2, 2, 0, device="cpu")
    finally:
        vx.VExpressPipeline.__init__ = orig
    assert torch.equal(got, ref)


def _call_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.set_num_threads(2)
    torch.manual_seed(1000 + rank)                 # every rank's default generator differs, as in a real launch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W.emulate_kernels()
    from v_express_amd import synth
    pipe = W.build_pipeline("cpu")
    cfg = cases.unet_cfg(cases.SMALL)
    F_ = 6
    inp = synth.synthetic_inputs(cfg, F_, 8, 8)
    lat = pipe(None, None, None, 64, 64, F_, 2, cases.GUIDANCE, context_frames=4, context_overlap=2,
               reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD, generator=None,
               reference_latents=inp["ref_latents"], kps_features=inp["kps_features"],
               audio_embeddings=inp["audio_embeddings"], decode=False)
    q.put((rank, lat.float().numpy().copy()))      # by value (see _worker)
    dist.barrier()
    dist.destroy_process_group()


def test_call_without_generator_gives_every_rank_the_same_clip():
    """with generator=None every rank draws its own noise; `__call__` must broadcast rank 0's draw, or the
    per-step all-gather mixes predictions of different clips.  Two gloo ranks, different default-generator seeds."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_call_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {r: torch.from_numpy(a) for r, a in (q.get(timeout=600) for _ in procs)}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.isfinite(results[0]).all() and torch.equal(results[0], results[1])


@pytest.mark.parametrize("layout", ["new_attn", "old_attn"])
def test_checkpoint_files_load_into_identical_models_unets(emulated, tmp_path, layout):
    """SURVEY.md 8f rank 4 on the CPU (emulated kernels): the bodies of tests/test_gpu_checkpoints.py."""
    import ckpt_cases
    ckpt_cases.unet_and_reference_net_from_files(tmp_path, layout, "cpu")


def test_checkpoint_files_load_into_identical_models_vae_guider_projection(emulated, tmp_path):
    import ckpt_cases
    ckpt_cases.vae_guider_and_audio_projection_from_files(tmp_path, "cpu")


def test_one_rank_group_forced_through_the_collective_path_equals_the_sequential_loop(emulated):
    """`DistContext(force=True)` on a ONE-rank group: the per-timestep all-gather, the broadcast and a sub-group all-to-all
    run as real collectives (what tools/rccl_world1_probe.py does on the one GPU that exists, over RCCL) and must change
    nothing."""
    import torch.distributed as dist
    from v_express_amd.distributed import CommTimer, DistContext, FrameShard
    F, cf, co = 14, 8, 2
    ref = W.run(F, cf, co, 2, 0, device="cpu")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        dc = DistContext(0, 1, None, force=True)
        assert dc.enabled and dc.backend == "gloo" and not DistContext().enabled
        t = torch.randn(3, 5)
        assert torch.equal(dc.broadcast(t.clone()), t) and torch.equal(dc.all_gather_units(t, 3)[0], t)
        fs = FrameShard(0, 1, dist.new_group([0]))
        x = torch.randn(2 * 4, 16, 8).to(torch.bfloat16)
        assert torch.equal(fs.to_frame_shard(fs.to_pixel_shard(x, 2, 4), 2, 4), x)
        pipe = W.build_pipeline("cpu")
        pipe.dist = dc
        with CommTimer() as tm:
            lat = W._run(pipe, pipe.denoising_unet, pipe.reference_net, pipe.scheduler, cases.unet_cfg(cases.SMALL),
                         F, cf, co, 2, 0, 16, "cpu")
        assert pipe.last_schedule["kind"] == "whole units" and tm.summary()["all_gather"]["calls"] == 2
        assert torch.equal(lat, ref)
    finally:
        dist.destroy_process_group()
