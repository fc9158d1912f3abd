"""Tier-A harness: run the reference's OWN modules/pipeline (imported unmodified, see ref_import.py) on the
synthetic weights/inputs of v_express_amd.synth.  TEST INFRASTRUCTURE; dev container only."""
import torch

import ref_import as R
from v_express_amd import synth


def build_reference_unets(cfg: synth.UNetConfig, sd3=None, sd2=None):
    modules, _ = R.import_reference()
    c = dict(R.SD15_UNET_CONFIG)
    c["block_out_channels"] = list(cfg.block_out_channels)
    refnet = modules.UNet2DConditionModel.from_config(dict(c))
    c3 = dict(c)
    c3["down_block_types"] = ["CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"]
    c3["up_block_types"] = ["UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"]
    c3["mid_block_type"] = "UNetMidBlock3DCrossAttn"
    unet = modules.UNet3DConditionModel.from_config(c3, **R.UNET_ADDITIONAL_KWARGS)
    unet.load_state_dict(sd3 if sd3 is not None else synth.unet3d_state_dict(cfg), strict=True)
    refnet.load_state_dict(sd2 if sd2 is not None else synth.refnet_state_dict(cfg), strict=True)
    return unet.eval(), refnet.eval()


def build_reference_vae(vcfg: synth.VaeConfig, sdv=None):
    R.import_reference()
    import diffusers
    vae = diffusers.AutoencoderKL(block_out_channels=tuple(vcfg.block_out_channels),
                                  layers_per_block=vcfg.layers_per_block,
                                  latent_channels=vcfg.latent_channels, norm_num_groups=vcfg.norm_num_groups)
    sdv = sdv if sdv is not None else synth.vae_decoder_state_dict(vcfg)
    missing, unexpected = vae.load_state_dict(sdv, strict=False)
    assert not unexpected and all(k.startswith(("encoder.", "quant_conv.")) for k in missing), (missing, unexpected)
    return vae.eval()


@torch.no_grad()
def reference_unet_forward(unet, refnet, inputs, t, w_ref, w_aud, frames=None):
    """ReferenceNet write -> reader.update(cfg) -> patched UNet3D forward, exactly as
    pipelines/v_express_pipeline.py:451-466,502-509,541-547 wires them.  Returns (pred, banks_by_order)."""
    modules, _ = R.import_reference()
    writer = modules.ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write",
                                               batch_size=1, fusion_blocks="full")
    reader = modules.ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", batch_size=1,
                                               fusion_blocks="full", reference_attention_weight=w_ref,
                                               audio_attention_weight=w_aud)
    ehs0 = torch.zeros(1, 1, 768)
    refnet(inputs["ref_latents"], timestep=0, encoder_hidden_states=ehs0, return_dict=False)
    reader.update(writer, True, dtype=torch.float32)
    lat = inputs["latents"]
    kps = inputs["kps_features"]
    aud = inputs["audio_embeddings"]
    if frames is not None:
        lat, kps, aud = lat[:, :, frames], kps[:, :, frames], aud[:, frames]
    x = lat.repeat(2, 1, 1, 1, 1)
    ehs = aud.reshape(-1, aud.shape[-2], aud.shape[-1])
    out = unet(x, t, encoder_hidden_states=ehs, kps_features=kps, return_dict=False)[0]
    banks = {}
    for name, m in refnet.named_modules():
        if hasattr(m, "bank") and name.endswith("transformer_blocks.0") and len(m.bank):
            banks[name[: -len(".transformer_blocks.0")]] = m.bank[0].clone()
    reader.clear()
    writer.clear()
    return out, banks


@torch.no_grad()
def reference_pipeline_run(unet, refnet, vae, inputs, num_frames, steps, guidance, ctx_frames, ctx_overlap,
                           w_ref, w_aud, height, width, decode=True, strength=1.0, start_latents=None,
                           on_step=None):
    """Run the reference's own VExpressPipeline.mean_overlap loop (+decode_latents) unmodified; only the
    out-of-scope prologue (VAE-encode of the reference image, kps guider, wav2vec2/audio projection —
    SURVEY.md §2 rows 13-15) is replaced by the synthetic tensors."""
    _, pipelines = R.import_reference()
    import diffusers
    sched = diffusers.DDIMScheduler(**R.NOISE_SCHEDULER_KWARGS)
    pipe = pipelines.VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet, v_kps_guider=None,
                                      audio_processor=None, audio_encoder=None, audio_projection=None,
                                      scheduler=sched)
    pipe.prepare_reference_latent = lambda *a, **k: inputs["ref_latents"]
    pipe.prepare_kps_feature = lambda *a, **k: inputs["kps_features"]
    pipe.prepare_audio_embeddings = lambda *a, **k: inputs["audio_embeddings"]
    # strength < 1 makes the reference's own get_timesteps (:334-341) start at a later timestep: with `start_latents` =
    # the latents after step k-1 this resumes an interrupted run bit-identically (every step is deterministic)
    lat0 = inputs["latents"] if start_latents is None else start_latents
    pipe.prepare_latents = lambda *a, **k: lat0.clone()
    trace = []
    if not decode:
        pipe.decode_latents = lambda latents: latents
    video = pipe(reference_image=None, kps_images=None, audio_waveform=None, width=width, height=height,
                 video_length=num_frames, num_inference_steps=steps, guidance_scale=guidance,
                 context_frames=ctx_frames, context_overlap=ctx_overlap, reference_attention_weight=w_ref,
                 audio_attention_weight=w_aud,
                 strength=strength,
                 callback=lambda i, t, l: (trace.append(l.clone()), on_step and on_step(i, t, l)))
    return video, trace
