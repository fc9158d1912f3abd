"""Oracle restatement of the sd-vae-ft-mse decode used by VExpressPipeline.decode_latents
(pipelines/v_express_pipeline.py:152-166) — TEST INFRASTRUCTURE.

The arithmetic is diffusers==0.29.2 `AutoencoderKL.decode` (absent third-party dependency; restated
from its published behaviour): post_quant_conv 1x1 -> conv_in 3x3 -> mid
(resnet, single-head attention with GroupNorm pre-norm + residual, resnet) -> 4 up blocks of 3 resnets
(nearest x2 + conv3x3 after the first three) -> GroupNorm(eps 1e-6) -> SiLU -> conv_out 3x3.
"""
import torch
import torch.nn.functional as F

from . import leaf as L
from .config import VaeConfig


def _vae_attention(w, p, x, groups):
    B, C, H, W = x.shape
    res = x
    h = x.view(B, C, H * W)
    h = F.group_norm(h, groups, w[p + ".group_norm.weight"], w[p + ".group_norm.bias"], 1e-6).transpose(1, 2)
    o = L.attention(w, p, h, h, heads=1)
    return o.transpose(1, 2).reshape(B, C, H, W) + res


def vae_decode(w, cfg: VaeConfig, z):
    """z [n,4,h,w] (already divided by the scaling factor) -> [n,3,8h,8w]."""
    g = cfg.norm_num_groups
    x = L.conv2d(w, "post_quant_conv", z, padding=0)
    x = L.conv2d(w, "decoder.conv_in", x)
    x = L.resnet(w, "decoder.mid_block.resnets.0", x, None, g, 1e-6)
    x = _vae_attention(w, "decoder.mid_block.attentions.0", x, g)
    x = L.resnet(w, "decoder.mid_block.resnets.1", x, None, g, 1e-6)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            x = L.resnet(w, f"decoder.up_blocks.{i}.resnets.{j}", x, None, g, 1e-6)
        if i != n - 1:
            x = L.upsample(w, f"decoder.up_blocks.{i}.upsamplers.0", x)
    x = F.silu(F.group_norm(x, g, w["decoder.conv_norm_out.weight"], w["decoder.conv_norm_out.bias"], 1e-6))
    return L.conv2d(w, "decoder.conv_out", x)


def decode_latents(w, cfg: VaeConfig, latents):
    """VExpressPipeline.decode_latents (pipelines/v_express_pipeline.py:152-166):
    latents [1,4,F,h,w] -> video [1,3,F,8h,8w] in [0,1], frame by frame."""
    b, c, f, h, wd = latents.shape
    lat = (latents / cfg.scaling_factor).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, wd)
    frames = []
    for i in range(lat.shape[0]):
        img = vae_decode(w, cfg, lat[i:i + 1])
        frames.append((img / 2 + 0.5).clamp(0, 1).float())
    video = torch.cat(frames)
    return video.reshape(b, f, *video.shape[1:]).permute(0, 2, 1, 3, 4)
