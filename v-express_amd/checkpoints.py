"""Weight ingestion for real V-Express checkpoints — host logic only.

Mirrors the reference's loaders and load ORDER (inference.py:77-129): ReferenceNet `strict=False`; denoising UNet
`strict=False` followed by the motion-module file `strict=False` (two partial state dicts into one model);
VKpsGuider and AudioProjection strict; VAE in the diffusers directory format.  The device re-layouts (fused QKV,
NHWC conv weights, GEGLU interleave, bf16) happen lazily at first use (weights.py), so what is loaded here are the
reference's own tensors under the reference's own keys.

`get_denoising_unet_state_dict` restates the legacy key remaps of train.py:122-161 ("old_attn", "moore_pretrained",
"new_attn") for checkpoints that predate the attn1_5 / norm1_5 reference-attention branch.
"""
import copy
import json
import os

import torch


def _load_file(path):
    """.bin / .pth / .pt through torch.load (CPU), .safetensors through safetensors."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu")


def get_denoising_unet_state_dict(old_state_dict, state_dict_type):
    """train.py:122-161.  `old_attn`: attn1/norm1 tensors are duplicated under attn1_5/norm1_5 and the audio
    cross-attention weights stored under `attn2.processor.to_*_aud` replace `attn2.to_*`; `moore_pretrained`: only
    the attn1 -> attn1_5 / norm1 -> norm1_5 duplication; `new_attn`: unchanged."""
    new = copy.deepcopy(old_state_dict)
    if state_dict_type == "old_attn":
        for name in old_state_dict.keys():
            if "norm1" in name:
                new[name.replace("norm1", "norm1_5")] = old_state_dict[name]
            if "attn1" in name:
                new[name.replace("attn1", "attn1_5")] = old_state_dict[name]
            for part in ("to_q", "to_k", "to_v", "to_out"):
                if f"attn2.{part}" in name:
                    new[name] = old_state_dict[name.replace(f"attn2.{part}", f"attn2.processor.{part}_aud")]
    elif state_dict_type == "moore_pretrained":
        for name in old_state_dict.keys():
            if "norm1" in name:
                new[name.replace("norm1", "norm1_5")] = old_state_dict[name]
            if "attn1" in name:
                new[name.replace("attn1", "attn1_5")] = old_state_dict[name]
    elif state_dict_type == "new_attn":
        pass
    else:
        raise ValueError(f'The state_dict_type {state_dict_type} is not supported. '
                         f'Only support "moore_pretrained", "old_attn", and "new_attn".')
    return new


def load_reference_net(unet_config_path, reference_net_path, dtype=torch.bfloat16, device="cuda"):
    """inference.py:77-81."""
    from .unet_2d import UNet2DConditionModel
    net = UNet2DConditionModel.from_config(unet_config_path).to(dtype=dtype, device=device)
    net.load_state_dict(_load_file(reference_net_path), strict=False)
    return net


def load_denoising_unet(unet_additional_kwargs, unet_config_path, denoising_unet_path, motion_module_path,
                        dtype=torch.bfloat16, device="cuda", state_dict_type="new_attn"):
    """inference.py:84-96: `unet_additional_kwargs` = the dict under that key of inference_v2.yaml (or a path to the
    yaml).  The motion-module file is loaded second, on top, exactly like the reference."""
    from .unet_3d import UNet3DConditionModel
    if isinstance(unet_additional_kwargs, str):
        import yaml
        with open(unet_additional_kwargs) as f:
            unet_additional_kwargs = yaml.safe_load(f)["unet_additional_kwargs"]
    unet = UNet3DConditionModel.from_config_2d(unet_config_path, unet_additional_kwargs=unet_additional_kwargs)
    unet = unet.to(dtype=dtype, device=device)
    unet.load_state_dict(get_denoising_unet_state_dict(_load_file(denoising_unet_path), state_dict_type), strict=False)
    if motion_module_path:
        unet.load_state_dict(_load_file(motion_module_path), strict=False)
    return unet


def load_v_kps_guider(v_kps_guider_path, dtype=torch.bfloat16, device="cuda"):
    """inference.py:99-103."""
    from .prologue import VKpsGuider
    m = VKpsGuider(320, block_out_channels=(16, 32, 96, 256)).to(dtype=dtype, device=device)
    m.load_state_dict(_load_file(v_kps_guider_path))
    return m


def load_audio_projection(audio_projection_path, dtype=torch.bfloat16, device="cuda", inp_dim=768, mid_dim=768,
                          out_dim=768, inp_seq_len=10, out_seq_len=5):
    """inference.py:106-129 (defaults = num_pad_audio_frames 2, cross_attention_dim 768)."""
    from .prologue import AudioProjection
    m = AudioProjection(dim=mid_dim, depth=4, dim_head=64, heads=12, num_queries=out_seq_len, embedding_dim=inp_dim,
                        output_dim=out_dim, ff_mult=4, max_seq_len=inp_seq_len).to(dtype=dtype, device=device)
    m.load_state_dict(_load_file(audio_projection_path))
    return m


def load_audio_encoder(audio_encoder_path, dtype=torch.bfloat16, device="cuda"):
    """inference.py:165-166: `Wav2Vec2Model.from_pretrained(path)` + `Wav2Vec2Processor.from_pretrained(path)` ->
    (encoder on the HIP kernels, waveform processor)."""
    from .wav2vec2 import Wav2Vec2Model, WaveformProcessor
    proc = WaveformProcessor()
    pc = os.path.join(audio_encoder_path, "preprocessor_config.json")
    if os.path.exists(pc):
        with open(pc) as f:
            cd = json.load(f)
        proc = WaveformProcessor(cd.get("sampling_rate", 16000), cd.get("do_normalize", True))
    return Wav2Vec2Model.from_pretrained(audio_encoder_path, dtype=dtype, device=device), proc


def load_vae(vae_dir, dtype=torch.bfloat16, device="cuda"):
    """AutoencoderKL.from_pretrained(vae_path) (inference.py:162) for the diffusers directory layout:
    `config.json` + `diffusion_pytorch_model.safetensors` (or `.bin`)."""
    from .synth import VaeConfig
    from .vae import AutoencoderKL
    with open(os.path.join(vae_dir, "config.json")) as f:
        cd = json.load(f)
    cfg = VaeConfig(latent_channels=cd.get("latent_channels", 4), out_channels=cd.get("out_channels", 3),
                    block_out_channels=tuple(cd.get("block_out_channels", (128, 256, 512, 512))),
                    layers_per_block=cd.get("layers_per_block", 2), norm_num_groups=cd.get("norm_num_groups", 32),
                    scaling_factor=cd.get("scaling_factor", 0.18215))
    for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"):
        path = os.path.join(vae_dir, name)
        if os.path.exists(path):
            sd = _load_file(path)
            break
    else:
        raise FileNotFoundError(f"no diffusion_pytorch_model.(safetensors|bin) under {vae_dir}")
    sd = {convert_vae_attention_key(k): v for k, v in sd.items()}
    vae = AutoencoderKL(cfg).to(dtype=dtype, device=device)
    vae.load_state_dict(sd)
    return vae


_VAE_ATTN_OLD = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def convert_vae_attention_key(key):
    """sd-vae-ft-mse was published with the pre-0.15 attention names (`query/key/value/proj_attn`); diffusers renames
    them to `to_q/to_k/to_v/to_out.0` at load time (`_convert_deprecated_attention_blocks`).  Same mapping here."""
    for old, new in _VAE_ATTN_OLD.items():
        tag = f".attentions.0.{old}."
        if tag in key:
            return key.replace(tag, f".attentions.0.{new}.")
    return key
