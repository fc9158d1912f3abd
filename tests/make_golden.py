"""Generate tests/golden/*.pt by running the REFERENCE ITSELF (imported unmodified over the diffusers
stand-in) on the seeded synthetic weights/inputs.  Run in the dev container:  python tests/make_golden.py
The GPU box has no /root/reference; it checks oracle/ and the HIP path against these committed outputs."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import cases  # noqa: E402
import refharness as H  # noqa: E402
from v_express_amd import synth  # noqa: E402


def prologue(out):
    """Reference outputs of the once-per-clip prologue and the median filter."""
    import types
    import ref_import as R
    modules, _ = R.import_reference()
    import diffusers   # the stand-in (on sys.path after import_reference)
    U = R.import_reference_utils()
    from pipelines.v_express_pipeline import VExpressPipeline
    inp = cases.prologue_inputs()
    g = {}
    for tag, kw in (("small", cases.KPS_SMALL), ("full", {})):
        kcfg = synth.KpsGuiderConfig(**kw)
        ref = modules.VKpsGuider(kcfg.conditioning_embedding_channels, block_out_channels=kcfg.block_out_channels)
        ref.load_state_dict(synth.kps_guider_state_dict(kcfg), strict=True)
        with torch.no_grad():
            g[f"kps_{tag}"] = ref(inp["kps_images"]).clone()
    for tag, kw, key in (("small", cases.AUDIO_SMALL, "audio_windows_small"), ("full", {}, "audio_windows_full")):
        acfg = synth.AudioProjectionConfig(**kw)
        ref = modules.AudioProjection(dim=acfg.dim, depth=acfg.depth, dim_head=acfg.dim_head, heads=acfg.heads,
                                      num_queries=acfg.num_queries, embedding_dim=acfg.embedding_dim,
                                      output_dim=acfg.output_dim, ff_mult=acfg.ff_mult, max_seq_len=acfg.max_seq_len)
        ref.load_state_dict(synth.audio_projection_state_dict(acfg), strict=True)
        with torch.no_grad():
            g[f"audio_{tag}"] = ref(inp[key]).clone()
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    vae = diffusers.AutoencoderKL(block_out_channels=vcfg.block_out_channels, layers_per_block=vcfg.layers_per_block,
                                  norm_num_groups=vcfg.norm_num_groups, latent_channels=vcfg.latent_channels)
    vae.load_state_dict(synth.vae_encoder_state_dict(vcfg), strict=False)
    with torch.no_grad():
        g["vae_mean"] = vae.encode(inp["ref_image"]).latent_dist.mean.clone()
    g["median"] = U.median_filter_3d(inp["video"], 3, "cpu").clone()
    g["median_u8"] = torch.from_numpy((g["median"].permute(1, 2, 3, 0) * 255).numpy().astype("uint8"))
    stub = types.SimpleNamespace(
        audio_processor=lambda wav, return_tensors, sampling_rate: {"input_values": wav},
        audio_encoder=lambda wav: types.SimpleNamespace(last_hidden_state=inp["wav2vec_states"]),
        audio_projection=lambda x: x, device="cpu", dtype=torch.float32)
    g["audio_windows_F7"] = VExpressPipeline.prepare_audio_embeddings(stub, torch.zeros(1, 16), 7, 2, False)[0].clone()
    torch.save(g, os.path.join(out, "prologue.pt"))
    print("prologue", {k: tuple(v.shape) for k, v in g.items()})


def wav2vec2(out):
    """Outputs of the installed transformers Wav2Vec2Model (the third-party module the reference calls at
    pipelines/v_express_pipeline.py:377) on the seeded synthetic weights and waveform."""
    g = {}
    for tag, (kw, samples) in cases.W2V_CASES.items():
        cfg = synth.Wav2Vec2Config(**kw)
        hf = cases.hf_wav2vec2(cfg, synth.wav2vec2_state_dict(cfg))
        with torch.no_grad():
            g[tag] = hf(cases.waveform(samples)).last_hidden_state.clone()
    import transformers
    g["transformers_version"] = transformers.__version__
    torch.save(g, os.path.join(out, "wav2vec2.pt"))
    print("wav2vec2", {k: (tuple(v.shape) if hasattr(v, "shape") else v) for k, v in g.items()})


def nocfg(out):
    """guidance_scale = 1.0: the reference loop without classifier-free guidance (batch of 1)."""
    cfg = cases.unet_cfg(cases.SMALL)
    unet, refnet = H.build_reference_unets(cfg)
    vae = H.build_reference_vae(synth.VaeConfig(**cases.SMALL_VAE))
    name, F, cf, co, steps = cases.NOCFG_CASE
    inp = cases.cond_only(synth.synthetic_inputs(cfg, F, 8, 8))
    video, trace = H.reference_pipeline_run(unet, refnet, vae, inp, F, steps, 1.0, cf, co, cases.W_REF, cases.W_AUD,
                                            64, 64)
    torch.save(dict(latents=trace[-1].clone(), latents_step0=trace[0].clone(), video_f16=video.to(torch.float16)),
               os.path.join(out, f"pipeline_{name}.pt"))
    print(name, video.shape, float(video.mean()))


def fullsize(out):
    """BASELINE.json configs[1] through the reference ITSELF: SD-1.5 widths, 64x64 latents (512x512), one 16-frame
    window (context 16 / overlap 4), CFG 3.5, 25 DDIM steps, sd-vae-ft-mse-shaped decode - the reference's own
    `VExpressPipeline.__call__` loop (pipelines/v_express_pipeline.py:526-589) and `decode_latents` (:152-166) on the
    seeded synthetic weights, fp32 on the host cores (~35-40 min on 8 idle cores).  Stores the first UNet prediction,
    the latents after steps 0 / 4 / 12 / 24 (= final) and two decoded frames.  Resumable: the latents after every step
    go to a scratch checkpoint; a restart continues through the reference's own `strength` path (get_timesteps,
    :334-341), which starts the same loop at a later timestep."""
    import time
    import types
    cfg = cases.unet_cfg(cases.FULL)
    unet, refnet = H.build_reference_unets(cfg)
    vae = H.build_reference_vae(synth.VaeConfig())
    F, cf, co, steps = cases.FULLSIZE_CASE
    inp = synth.synthetic_inputs(cfg, F, 64, 64)
    ckpt_path = os.path.join(os.path.dirname(HERE), "gpurun_out", "fullsize_ckpt.pt")
    ck = torch.load(ckpt_path, weights_only=False) if os.path.exists(ckpt_path) else dict(done=0, keep={}, secs=0.0)
    orig = unet.forward

    def spy(*a, **k):
        o = orig(*a, **k)
        if "pred_step0" not in ck["keep"]:
            ck["keep"]["pred_step0"] = (o[0] if isinstance(o, tuple) else o.sample).clone()
        return o
    unet.forward = spy
    base = ck["done"]
    t_last = [time.time()]

    def on_step(i, t, lat):
        g = base + i
        if g in (0, 4, 12, steps - 1):
            ck["keep"]["latents" if g == steps - 1 else f"latents_step{g}"] = lat.clone()
        ck["done"], ck["last"] = g + 1, lat.clone()
        ck["secs"] += time.time() - t_last[0]
        t_last[0] = time.time()
        os.makedirs(os.path.dirname(ckpt_path), exist_ok=True)
        torch.save(ck, ckpt_path + ".tmp")
        os.replace(ckpt_path + ".tmp", ckpt_path)
        print(f"step {g} t={int(t)} done ({ck['secs']:.0f} s so far)", flush=True)
    if base < steps:
        H.reference_pipeline_run(unet, refnet, vae, inp, F, steps, cases.GUIDANCE, cf, co, cases.W_REF, cases.W_AUD,
                                 512, 512, decode=False, strength=(steps - base + 0.5) / steps,
                                 start_latents=ck.get("last"), on_step=on_step)
    g = dict(ck["keep"], loop_seconds=ck["secs"], threads=torch.get_num_threads())
    # decode_latents as the reference does it (v_express_pipeline.py:152-166), two frames kept
    _, pipelines = __import__("ref_import").import_reference()
    t0 = time.time()
    keep = list(cases.FULLSIZE_FRAMES)
    video = pipelines.VExpressPipeline.decode_latents(types.SimpleNamespace(vae=vae), g["latents"][:, :, keep])
    g["decode_seconds_per_frame"] = (time.time() - t0) / len(keep)
    g["video_frames"] = keep
    g["video_f16"] = video.to(torch.float16)
    torch.save(g, os.path.join(out, "fullsize_F16_512.pt"))
    print("fullsize", {k: (tuple(v.shape) if hasattr(v, "shape") else v) for k, v in g.items()})


def fullsize_F28(out, case=None, fname="fullsize_F28_512.pt"):
    """The sliding-window / merged-call path at the BENCHMARKED geometry (BASELINE.json configs[2] shape, shortened):
    the reference's own `VExpressPipeline.__call__` (pipelines/v_express_pipeline.py:526-589) at SD-1.5 widths, 64x64
    latents (512x512), F = 28 with context 16 / overlap 4 -> two windows [0..15], [12..27] sharing four frames, CFG 3.5,
    2 DDIM steps (4 CFG forwards of 16 frames, ~6 min on 8 cores).  Stores every window's first prediction (fp16) and
    the latents after each step (fp32): what the b = 4 merged UNet call and the overlap averaging at 64x64 latents are
    compared with on the GPU (tests/test_gpu_fullsize.py)."""
    import time
    cfg = cases.unet_cfg(cases.FULL)
    unet, refnet = H.build_reference_unets(cfg)
    vae = H.build_reference_vae(synth.VaeConfig(**cases.SMALL_VAE))     # only its scale factor (8) is used: no decode
    F, cf, co, steps = case or cases.FULLSIZE_F28_CASE
    inp = synth.synthetic_inputs(cfg, F, 64, 64)
    orig = unet.forward
    preds = []

    def spy(*a, **k):
        o = orig(*a, **k)
        preds.append((o[0] if isinstance(o, tuple) else o.sample).clone())
        print(f"unet call {len(preds)} done", flush=True)
        return o
    unet.forward = spy
    t0 = time.time()
    _, trace = H.reference_pipeline_run(unet, refnet, vae, inp, F, steps, cases.GUIDANCE, cf, co, cases.W_REF,
                                        cases.W_AUD, 512, 512, decode=False)
    from pipelines.context import uniform
    wins = list(uniform(step=0, num_frames=F, context_size=cf, context_stride=1, context_overlap=co,
                        closed_loop=False))
    per_step = len(preds) // steps
    first = torch.cat(preds[:per_step], dim=0)           # step 0: [n_windows * 2, 4, 16, 64, 64] in call order
    g = dict(windows=wins, pred_step0_f16=first.to(torch.float16), calls_per_step=per_step,
             pred_shapes=[tuple(p.shape) for p in preds[:per_step]],
             latents_step0=trace[0].clone(), latents=trace[-1].clone(), loop_seconds=time.time() - t0,
             threads=torch.get_num_threads())
    torch.save(g, os.path.join(out, fname))
    print(fname, {k: (tuple(v.shape) if hasattr(v, "shape") else v) for k, v in g.items()})


def fullsize_ctx24(out):
    """The reference's DEFAULT window geometry (inference.py:67-68: --context_frames 24 --context_overlap 4; the
    positional table of the motion modules holds 32 entries, inference_v2.yaml:21) at SD-1.5 widths and 64x64 latents:
    F = 44 -> windows [0..23] and [20..43] sharing four frames, CFG 3.5, 2 DDIM steps of the reference's own
    `VExpressPipeline.__call__` (4 CFG forwards of 24 frames, ~10 min on 8 cores).  Same contents as fullsize_F28."""
    return fullsize_F28(out, cases.FULLSIZE_CTX24_CASE, "fullsize_ctx24_F44_512.pt")


def fullsize_768(out):
    """BASELINE.json configs[4]'s geometry through the reference ITSELF: SD-1.5 widths, 96x96 latents (768x768), one
    4-frame window, CFG 3.5, 2 DDIM steps of `VExpressPipeline.__call__` (pipelines/v_express_pipeline.py:526-589) -
    2 CFG forwards of 4 frames (~6 min on 8 cores).  Stores the first prediction (fp16) and the latents after each step."""
    import time
    cfg = cases.unet_cfg(cases.FULL)
    unet, refnet = H.build_reference_unets(cfg)
    vae = H.build_reference_vae(synth.VaeConfig(**cases.SMALL_VAE))     # only its scale factor (8) is used: no decode
    F, cf, co, steps = cases.FULLSIZE_768_CASE
    inp = synth.synthetic_inputs(cfg, F, 96, 96)
    orig = unet.forward
    preds = []

    def spy(*a, **k):
        o = orig(*a, **k)
        preds.append((o[0] if isinstance(o, tuple) else o.sample).clone())
        print(f"unet call {len(preds)} done", flush=True)
        return o
    unet.forward = spy
    t0 = time.time()
    _, trace = H.reference_pipeline_run(unet, refnet, vae, inp, F, steps, cases.GUIDANCE, cf, co, cases.W_REF,
                                        cases.W_AUD, 768, 768, decode=False)
    g = dict(pred_step0_f16=preds[0].to(torch.float16), latents_step0=trace[0].clone(), latents=trace[-1].clone(),
             loop_seconds=time.time() - t0, threads=torch.get_num_threads())
    torch.save(g, os.path.join(out, "fullsize_768_F4.pt"))
    print("fullsize_768", {k: (tuple(v.shape) if hasattr(v, "shape") else v) for k, v in g.items()})


def main():
    torch.set_num_threads(os.cpu_count())
    out = os.path.join(HERE, "golden")
    os.makedirs(out, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "prologue":
        return prologue(out)
    if len(sys.argv) > 1 and sys.argv[1] == "wav2vec2":
        return wav2vec2(out)
    if len(sys.argv) > 1 and sys.argv[1] == "nocfg":
        return nocfg(out)
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":      # ~40 min; not part of the default regeneration
        return fullsize(out)
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize_F28":  # ~6 min; not part of the default regeneration
        return fullsize_F28(out)
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize_ctx24":  # ~10 min; not part of the default regeneration
        return fullsize_ctx24(out)
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize_768":  # ~6 min; not part of the default regeneration
        return fullsize_768(out)
    prologue(out)
    wav2vec2(out)
    built = {}
    for name, (kw, F, h, w, t) in cases.FORWARD_CASES.items():
        key = tuple(kw["block_out_channels"])
        if key not in built:
            cfg = cases.unet_cfg(kw)
            built = {key: H.build_reference_unets(cfg)}        # keep only one model pair alive
        unet, refnet = built[key]
        inp = synth.synthetic_inputs(cases.unet_cfg(kw), F, h, w)
        pred, banks = H.reference_unet_forward(unet, refnet, inp, t, cases.W_REF, cases.W_AUD)
        g = dict(pred=pred.clone(),
                 bank_stats={k: torch.stack([v.mean(), v.abs().mean(), v.flatten()[::97].sum()])
                             for k, v in banks.items()})
        torch.save(g, os.path.join(out, f"forward_{name}.pt"))
        print(name, pred.shape, float(pred.std()))
    kw = cases.SMALL
    cfg = cases.unet_cfg(kw)
    unet, refnet = H.build_reference_unets(cfg)
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    vae = H.build_reference_vae(vcfg)
    for name, (F, cf, co, steps) in cases.PIPELINE_CASES.items():
        inp = synth.synthetic_inputs(cfg, F, 8, 8)
        video, trace = H.reference_pipeline_run(unet, refnet, vae, inp, F, steps, cases.GUIDANCE, cf, co,
                                                cases.W_REF, cases.W_AUD, 64, 64)
        g = dict(latents=trace[-1].clone(), latents_step0=trace[0].clone(),
                 video_f16=video.to(torch.float16))
        torch.save(g, os.path.join(out, f"pipeline_{name}.pt"))
        print(name, video.shape, float(video.mean()))
    nocfg(out)
    # window lists straight from the reference's pipelines/context.py
    from pipelines.context import uniform
    wins = {}
    for (F, cs, co) in [(16, 16, 4), (64, 16, 4), (124, 16, 4), (128, 16, 4), (100, 24, 4), (924, 24, 4), (11, 4, 2),
                        (7, 8, 2), (40, 24, 4)]:
        wins[(F, cs, co)] = list(uniform(step=0, num_frames=F, context_size=cs, context_stride=1,
                                         context_overlap=co, closed_loop=False))
    torch.save(wins, os.path.join(out, "windows.pt"))
    # DDIM constants straight from the stand-in scheduler the reference pipeline drives
    import diffusers
    import ref_import as R
    s = diffusers.DDIMScheduler(**R.NOISE_SCHEDULER_KWARGS)
    s.set_timesteps(25)
    torch.save(dict(timesteps=s.timesteps.clone(), alphas_cumprod=s.alphas_cumprod.clone()),
               os.path.join(out, "ddim.pt"))


if __name__ == "__main__":
    main()
