"""Oracle restatements of the once-per-clip prologue and the post-processing step — TEST INFRASTRUCTURE.

SURVEY.md §8f ranks 2 and 3: VKpsGuider, AudioProjection, the audio-window construction, the VAE *encode* of the
reference image (diffusers AutoencoderKL.encode, absent third-party dependency, restated like the decoder) and
`median_filter_3d` + uint8 packing.  Plain fp32 PyTorch over flat weight dicts with the reference's key names; each
function cites the reference file:line it follows.  Pinned against the reference itself (modules imported unmodified,
tests/test_oracle_vs_reference.py) and through tests/golden/prologue_*.pt.
"""
import math

import torch
import torch.nn.functional as F

from . import leaf as L
from .config import VaeConfig


def kps_guider(w, x, n_blocks=6):
    """VKpsGuider.forward (modules/v_kps_guider.py:35-45) on per-frame images x [(b f), 3, H, W]:
    conv_in -> SiLU -> [conv3x3 -> SiLU -> conv3x3 stride 2 -> SiLU] x 3 -> conv_out (InflatedConv3d == per-frame
    conv2d, modules/resnet.py:9-17)."""
    h = F.silu(L.conv2d(w, "conv_in", x))
    for i in range(n_blocks):
        h = F.silu(L.conv2d(w, f"blocks.{i}", h, stride=2 if i % 2 else 1))
    return L.conv2d(w, "conv_out", h)


def _perceiver_attention(w, p, x, latents, heads):
    """PerceiverAttention.forward (modules/audio_projection.py:48-85)."""
    x = L.layer_norm(w, p + ".norm1", x)
    latents = L.layer_norm(w, p + ".norm2", latents)
    b, l, _ = latents.shape
    q = F.linear(latents, w[p + ".to_q.weight"])
    k, v = F.linear(torch.cat((x, latents), dim=-2), w[p + ".to_kv.weight"]).chunk(2, dim=-1)
    d = q.shape[-1] // heads

    def split(t):
        return t.view(b, t.shape[1], heads, d).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    scale = 1 / math.sqrt(math.sqrt(d))
    weight = torch.softmax(((q * scale) @ (k * scale).transpose(-2, -1)).float(), dim=-1)
    out = (weight @ v).permute(0, 2, 1, 3).reshape(b, l, -1)
    return F.linear(out, w[p + ".to_out.weight"])


def audio_projection(w, x, depth, heads):
    """AudioProjection.forward (modules/audio_projection.py:130-150), num_latents_mean_pooled = 0.
    x [F, n, 768] -> [F, num_queries, output_dim]."""
    n = x.shape[1]
    x = x + w["pos_emb.weight"][:n]
    latents = w["latents"].repeat(x.shape[0], 1, 1)
    x = L.linear(w, "proj_in", x)
    for i in range(depth):
        latents = _perceiver_attention(w, f"layers.{i}.0", x, latents, heads) + latents
        f = f"layers.{i}.1"
        h = L.layer_norm(w, f + ".0", latents)
        h = F.linear(F.gelu(F.linear(h, w[f + ".1.weight"])), w[f + ".3.weight"])
        latents = h + latents
    return L.layer_norm(w, "norm_out", L.linear(w, "proj_out", latents))


def audio_windows(audio_embeddings, video_length, num_pad_audio_frames):
    """VExpressPipeline.prepare_audio_embeddings, :381-401: [1, T, d] wav2vec2 states -> linear interpolation to
    2*video_length, zero padding of 2*num_pad on both sides, one window of 2*(2*num_pad+1) rows per frame."""
    emb = F.interpolate(audio_embeddings.float().permute(0, 2, 1), size=2 * video_length, mode="linear")[0].permute(1, 0)
    pad = torch.zeros_like(emb)[:2 * num_pad_audio_frames]
    emb = torch.cat([pad, emb, pad], dim=0)
    return torch.stack([emb[2 * i:2 * (i + 2 * num_pad_audio_frames + 1)] for i in range(video_length)], dim=0)


def _vae_attention(w, p, x, groups):
    B, C, H, W = x.shape
    h = F.group_norm(x.view(B, C, H * W), groups, w[p + ".group_norm.weight"], w[p + ".group_norm.bias"], 1e-6)
    o = L.attention(w, p, h.transpose(1, 2), h.transpose(1, 2), heads=1)
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def vae_encode_mean(w, cfg: VaeConfig, x):
    """diffusers AutoencoderKL.encode(x).latent_dist.mean as used by prepare_reference_latent
    (pipelines/v_express_pipeline.py:343-348): conv_in -> 4 down blocks of 2 resnets (+ pad (0,1,0,1) + conv3x3
    stride 2 after the first three) -> mid (resnet, attention, resnet) -> GroupNorm(eps 1e-6) -> SiLU -> conv_out
    (8 channels) -> quant_conv 1x1; the mean is the first half of the channels.  x [n,3,H,W] in [-1,1]."""
    g = cfg.norm_num_groups
    h = L.conv2d(w, "encoder.conv_in", x)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = L.resnet(w, f"encoder.down_blocks.{i}.resnets.{j}", h, None, g, 1e-6)
        if i != n - 1:
            h = L.conv2d(w, f"encoder.down_blocks.{i}.downsamplers.0.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = L.resnet(w, "encoder.mid_block.resnets.0", h, None, g, 1e-6)
    h = _vae_attention(w, "encoder.mid_block.attentions.0", h, g)
    h = L.resnet(w, "encoder.mid_block.resnets.1", h, None, g, 1e-6)
    h = F.silu(F.group_norm(h, g, w["encoder.conv_norm_out.weight"], w["encoder.conv_norm_out.bias"], 1e-6))
    moments = L.conv2d(w, "quant_conv", L.conv2d(w, "encoder.conv_out", h), padding=0)
    return moments[:, :cfg.latent_channels]


def median_filter_3d(video, kernel_size=3):
    """pipelines/utils.py:46-61 without the per-frame device hops: video [C, F, H, W] -> same shape."""
    c, f, h, w = video.shape
    p = kernel_size // 2
    v = F.pad(video[None], (p, p, p, p, p, p), mode="reflect")[0]
    out = []
    for i in range(f):
        seg = v[:, i:i + kernel_size].unfold(2, kernel_size, 1).unfold(3, kernel_size, 1)
        seg = seg.permute(0, 2, 3, 1, 4, 5).reshape(c, h, w, -1)
        out.append(torch.median(seg, dim=-1)[0])
    return torch.stack(out, dim=1)


def frames_uint8(video):
    """save_video, pipelines/utils.py:70-73: [C, F, H, W] float in [0,1] -> uint8 [F, H, W, C] (truncation)."""
    return (video.permute(1, 2, 3, 0) * 255).numpy().astype("uint8")
