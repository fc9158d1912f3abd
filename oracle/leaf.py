"""Leaf arithmetic of the oracle (TEST INFRASTRUCTURE) — restated diffusers==0.29.2 leaves.

The reference delegates this math to diffusers (not vendored, not installable here); each function
names the reference call site that reaches it and the diffusers class it restates.
All functions take a flat weight dict `w` (reference state_dict key names) and a key prefix `p`.
"""
import math

import torch
import torch.nn.functional as F


def linear(w, p, x):
    return F.linear(x, w[p + ".weight"], w.get(p + ".bias"))


def conv2d(w, p, x, stride=1, padding=1):
    """InflatedConv3d == per-frame nn.Conv2d (modules/resnet.py:9-17)."""
    return F.conv2d(x, w[p + ".weight"], w.get(p + ".bias"), stride=stride, padding=padding)


def group_norm(w, p, x, groups, eps):
    """InflatedGroupNorm == per-frame nn.GroupNorm (modules/resnet.py:20-28)."""
    return F.group_norm(x, groups, w[p + ".weight"], w[p + ".bias"], eps)


def layer_norm(w, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w[p + ".weight"], w[p + ".bias"], eps)


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0):
    """diffusers Timesteps (called at modules/unet_3d.py:464): fp32 sinusoid, [cos | sin] when flipped."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    ang = t[:, None].float() * torch.exp(exponent)[None]
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def time_embedding(w, t, dim0):
    """Timesteps + TimestepEmbedding (modules/unet_3d.py:464-470): Linear, SiLU, Linear."""
    e = timestep_embedding(t, dim0)
    e = linear(w, "time_embedding.linear_1", e)
    return linear(w, "time_embedding.linear_2", F.silu(e))


USE_SDPA = [False]


def attention(w, p, x, ctx, heads):
    """diffusers Attention + AttnProcessor2_0 (instantiated modules/attention.py:321-360,
    modules/motion_module.py:280-290): q=xWq, k=cWk, v=cWv (no bias), softmax(qk^T/sqrt(d))v, Wo+bias."""
    B, N, C = x.shape
    q = F.linear(x, w[p + ".to_q.weight"], w.get(p + ".to_q.bias"))
    k = F.linear(ctx, w[p + ".to_k.weight"], w.get(p + ".to_k.bias"))
    v = F.linear(ctx, w[p + ".to_v.weight"], w.get(p + ".to_v.bias"))
    d = q.shape[-1] // heads
    q = q.view(B, -1, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    if USE_SDPA[0]:
        # what AttnProcessor2_0 itself calls (same value, no [N, N] score tensor: 2 GB per call at 4096 tokens);
        # bench.py's cpu_baseline leg times the oracle this way, the parity tests keep the explicit form below
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
    else:
        s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
        o = torch.matmul(torch.softmax(s, dim=-1), v)
    o = o.transpose(1, 2).reshape(B, -1, heads * d)
    return linear(w, p + ".to_out.0", o)


def feed_forward(w, p, x):
    """diffusers FeedForward(activation_fn='geglu') (modules/attention.py:375, motion_module.py:233):
    h, g = Linear(C->8C)(x).chunk(2); Linear(4C->C)(h * gelu_erf(g))."""
    h, g = linear(w, p + ".net.0.proj", x).chunk(2, dim=-1)
    return linear(w, p + ".net.2", h * F.gelu(g))


def resnet(w, p, x, temb, groups, eps):
    """ResnetBlock3D.forward (modules/resnet.py:217-251) on (b f) c h w frames; `temb` is already
    broadcast to one row per frame.  Also diffusers ResnetBlock2D (ReferenceNet, VAE: temb=None)."""
    h = F.silu(group_norm(w, p + ".norm1", x, groups, eps))
    h = conv2d(w, p + ".conv1", h)
    if temb is not None:
        h = h + linear(w, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(group_norm(w, p + ".norm2", h, groups, eps))
    h = conv2d(w, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in w:
        x = conv2d(w, p + ".conv_shortcut", x, padding=0)
    return x + h


def upsample(w, p, x):
    """Upsample3D / Upsample2D: nearest x2 in H,W then conv3x3 (modules/resnet.py:53-90)."""
    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    return conv2d(w, p + ".conv", x)


def downsample(w, p, x):
    """Downsample3D / Downsample2D: conv3x3 stride 2 pad 1 (modules/resnet.py:106-118)."""
    return conv2d(w, p + ".conv", x, stride=2, padding=1)
