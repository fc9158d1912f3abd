"""Bodies of the checkpoint-ingestion tests (tests/test_gpu_checkpoints.py on the GPU; tests/test_host_emulated.py
over emulated kernels on the CPU).  TEST INFRASTRUCTURE.

Checkpoint ingestion ON THE DEVICE: synthetic checkpoint files in the reference's formats
(`.bin` through torch.save, `.safetensors`, the diffusers VAE directory with the deprecated attention key names, the
legacy "old_attn" denoising-UNet layout) are written to tmp_path, loaded through `checkpoints.load_*` in the reference's
load order (inference.py:77-129: ReferenceNet strict=False; denoising UNet strict=False, then the motion-module file on
top; guider / audio projection strict; VAE directory) and every loaded model's forward must equal, BIT FOR BIT, the
model that received the same tensors through `load_state_dict` directly."""
import contextlib
import json
import os

import pytest
import torch

import cases
from v_express_amd import module_base

def _unet_config_json(path, kw):
    cfg = cases.unet_cfg(kw)
    cd = dict(in_channels=4, out_channels=4, block_out_channels=list(cfg.block_out_channels), layers_per_block=2,
              attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5,
              down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
              up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3, sample_size=64, act_fn="silu")
    path.write_text(json.dumps(cd))
    return cfg


def _to_old_attn(sd):
    """Inverse of train.py:122-161's "old_attn" remap: a checkpoint that predates attn1_5 / norm1_5 and keeps the audio
    cross-attention under `attn2.processor.to_*_aud`.  attn1_5 / norm1_5 then come out as copies of attn1 / norm1."""
    old = {}
    for k, v in sd.items():
        if "attn1_5" in k or "norm1_5" in k:
            continue
        hit = False
        for part in ("to_q", "to_k", "to_v", "to_out"):
            if f"attn2.{part}" in k:
                old[k.replace(f"attn2.{part}", f"attn2.processor.{part}_aud")] = v
                old[k] = torch.zeros_like(v)                # the stale text cross-attention weights of SD-1.5
                hit = True
        if not hit:
            old[k] = v
    return old


def unet_and_reference_net_from_files(tmp_path, layout, device):
    from safetensors.torch import save_file
    from v_express_amd import ReferenceAttentionControl, UNet2DConditionModel, UNet3DConditionModel, checkpoints, synth
    import ref_import as R
    cfg = _unet_config_json(tmp_path / "config.json", cases.SMALL)
    sd3, sd2 = synth.unet3d_state_dict(cfg), synth.refnet_state_dict(cfg)
    motion = {k: v for k, v in sd3.items() if "motion_module" in k}               # train.py:744-753
    spatial = {k: v for k, v in sd3.items() if "motion_module" not in k}
    if layout == "old_attn":
        spatial = _to_old_attn(spatial)
        remapped = checkpoints.get_denoising_unet_state_dict(spatial, "old_attn")
        want3 = {**{k: remapped[k] for k in sd3 if k in remapped}, **motion}
        assert torch.equal(want3["down_blocks.0.attentions.0.transformer_blocks.0.attn1_5.to_q.weight"],
                           sd3["down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight"])
    else:
        want3 = sd3
    torch.save(spatial, tmp_path / "denoising_unet.bin")
    save_file({k: v.contiguous() for k, v in motion.items()}, str(tmp_path / "motion_module.safetensors"))
    torch.save(sd2, tmp_path / "reference_net.bin")
    unet = checkpoints.load_denoising_unet(R.UNET_ADDITIONAL_KWARGS, str(tmp_path / "config.json"),
                                           str(tmp_path / "denoising_unet.bin"),
                                           str(tmp_path / "motion_module.safetensors"), state_dict_type=layout, device=device)
    refnet = checkpoints.load_reference_net(str(tmp_path / "config.json"), str(tmp_path / "reference_net.bin"), device=device)
    assert unet.device.type == device and unet.dtype == torch.bfloat16
    assert set(unet.state_dict()) >= set(sd3)                                     # every tensor of the model was filled
    unet_d = UNet3DConditionModel(cfg).to(device)
    unet_d.load_state_dict(want3, strict=True)
    refnet_d = UNet2DConditionModel(cfg).to(device)
    refnet_d.load_state_dict(sd2, strict=True)
    F, h, w, t = 4, 8, 8, 519
    inp = synth.synthetic_inputs(cfg, F, h, w)
    outs = []
    for u, r in ((unet, refnet), (unet_d, refnet_d)):
        writer = ReferenceAttentionControl(r, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
        reader = ReferenceAttentionControl(u, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                           reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD)
        r(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
        reader.update(writer, True)
        x = inp["latents"].repeat(2, 1, 1, 1, 1)
        ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
        outs.append(u(x, t, encoder_hidden_states=ehs, kps_features=inp["kps_features"], return_dict=False)[0].cpu())
        reader.clear()
        writer.clear()
    assert torch.isfinite(outs[0]).all() and outs[0].abs().max() > 0
    assert torch.equal(outs[0], outs[1]), "the file-loaded UNet3D / ReferenceNet differ from the load_state_dict ones"
    # the reference's default dtype (inference.py:44): a compute dtype since round 6 (the IEEE-half build of the library)
    r16 = checkpoints.load_reference_net(str(tmp_path / "config.json"), str(tmp_path / "reference_net.bin"),
                                         dtype=torch.float16, device=device)
    assert r16.dtype == torch.float16 and r16._elem == torch.float16


def vae_guider_and_audio_projection_from_files(tmp_path, device):
    from safetensors.torch import save_file
    from v_express_amd import AudioProjection, AutoencoderKL, VKpsGuider, checkpoints, synth
    # --- VAE: diffusers directory, sd-vae-ft-mse's pre-0.15 attention names (query / key / value / proj_attn)
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    sdv = {**synth.vae_decoder_state_dict(vcfg), **synth.vae_encoder_state_dict(vcfg)}
    new2old = {v: k for k, v in checkpoints._VAE_ATTN_OLD.items()}
    old = {}
    for k, v in sdv.items():
        for new, o in new2old.items():
            tag = f".attentions.0.{new}."
            if tag in k:
                k = k.replace(tag, f".attentions.0.{o}.")
                break
        old[k] = v.contiguous()
    assert any(".query." in k for k in old)
    d = tmp_path / "sd-vae"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(dict(block_out_channels=list(vcfg.block_out_channels), latent_channels=4,
                                                   layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)))
    save_file(old, str(d / "diffusion_pytorch_model.safetensors"))
    vae = checkpoints.load_vae(str(d), device=device)
    vae_d = AutoencoderKL(vcfg).to(device)
    vae_d.load_state_dict(sdv)
    inp = cases.prologue_inputs()
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(5))
    assert torch.equal(vae.decode(z).sample.cpu(), vae_d.decode(z).sample.cpu())
    assert torch.equal(vae.encode(inp["ref_image"]).latent_dist.mean.cpu(),
                       vae_d.encode(inp["ref_image"]).latent_dist.mean.cpu())
    # --- VKpsGuider / AudioProjection at the real inference.py:99-129 configurations, strict
    kcfg, acfg = synth.KpsGuiderConfig(), synth.AudioProjectionConfig(
        dim=768, depth=4, dim_head=64, heads=12, num_queries=5, embedding_dim=768, output_dim=768, ff_mult=4,
        max_seq_len=10)
    sdk, sda = synth.kps_guider_state_dict(kcfg), synth.audio_projection_state_dict(acfg)
    torch.save(sdk, tmp_path / "v_kps_guider.bin")
    torch.save(sda, tmp_path / "audio_projection.bin")
    g = checkpoints.load_v_kps_guider(str(tmp_path / "v_kps_guider.bin"), device=device)
    g_d = VKpsGuider(kcfg.conditioning_embedding_channels, block_out_channels=kcfg.block_out_channels).to(device)
    g_d.load_state_dict(sdk)
    assert torch.equal(g(inp["kps_images"]).cpu(), g_d(inp["kps_images"]).cpu())
    a = checkpoints.load_audio_projection(str(tmp_path / "audio_projection.bin"), device=device)
    a_d = AudioProjection(dim=768, depth=4, dim_head=64, heads=12, num_queries=5, embedding_dim=768, output_dim=768,
                          ff_mult=4, max_seq_len=10).to(device)
    a_d.load_state_dict(sda)
    assert torch.equal(a(inp["audio_windows_full"]).cpu(), a_d(inp["audio_windows_full"]).cpu())
    # strict loading like the reference (inference.py:101,127): a wrong file is refused at load time
    bad = dict(sdk)
    bad.pop("conv_out.bias")
    torch.save(bad, tmp_path / "bad.bin")
    with pytest.raises(RuntimeError):
        checkpoints.load_v_kps_guider(str(tmp_path / "bad.bin"), device=device)
    bad = dict(sda, **{"proj_in.weight": torch.zeros(3, 3)})
    torch.save(bad, tmp_path / "bad2.bin")
    with pytest.raises(RuntimeError):
        checkpoints.load_audio_projection(str(tmp_path / "bad2.bin"), device=device)
