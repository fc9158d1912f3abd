"""Oracle restatement of the two UNets on the hot path (TEST INFRASTRUCTURE).

* `unet3d_forward`  — UNet3DConditionModel.forward (modules/unet_3d.py:400-578) with every
  TemporalBasicTransformerBlock running the patched *read* branch
  (modules/mutual_self_attention.py:176-267) and every motion module
  (modules/motion_module.py:146-182,236-259,351-388).
* `refnet_banks`    — ReferenceNet = UNet2DConditionModel.forward (modules/unet_2d_condition.py:877-1313)
  with every BasicTransformerBlock running the patched *write* branch
  (modules/mutual_self_attention.py:145-174,269-284); returns the 16 bank tensors by block name.

Bank pairing: the reference pairs writer/reader blocks by a stable sort on channel width over DFS
order (mutual_self_attention.py:341-357); both UNets register children as down -> up -> mid, so the
pairing is name-identical.  We key banks by the block prefix.
"""
import torch
import torch.nn.functional as F

from . import leaf as L
from .config import UNetConfig


# ----------------------------------------------------------------------------- structure walk
def block_plan(cfg: UNetConfig):
    """Yields the (kind, prefix, ...) sequence both UNets share (unet_3d.py:112-227; App. B)."""
    ch = cfg.block_out_channels
    n = len(ch)
    plan = {"down": [], "mid": None, "up": []}
    out_c = ch[0]
    for i in range(n):
        in_c, out_c = out_c, ch[i]
        layers = []
        for j in range(cfg.layers_per_block):
            layers.append(dict(cin=in_c if j == 0 else out_c, cout=out_c))
        plan["down"].append(dict(prefix=f"down_blocks.{i}", layers=layers, attn=cfg.attn_levels[i],
                                 sampler=(i != n - 1), c=out_c))
    plan["mid"] = dict(prefix="mid_block", c=ch[-1])
    rev = list(reversed(ch))
    rev_attn = list(reversed(cfg.attn_levels))
    out_c = rev[0]
    for i in range(n):
        prev_out = out_c
        out_c = rev[i]
        in_c = rev[min(i + 1, n - 1)]
        layers = []
        nl = cfg.layers_per_block + 1
        for j in range(nl):
            skip = in_c if j == nl - 1 else out_c
            rin = prev_out if j == 0 else out_c
            layers.append(dict(cin=rin + skip, cout=out_c))
        plan["up"].append(dict(prefix=f"up_blocks.{i}", layers=layers, attn=rev_attn[i],
                               sampler=(i != n - 1), c=out_c))
    return plan


# ----------------------------------------------------------------------------- 3-D denoising UNet
def _spatial_transformer_read(w, p, x, f, ehs, bank, cfg, w_ref, w_aud):
    """Transformer3DModel.forward (modules/transformer_3d.py:103-169) + patched read branch
    (modules/mutual_self_attention.py:176-267).  x: [(b f), C, H, W]; ehs: [(b f), 5, 768];
    bank: [b, HW, C] (CFG: row 0 zeros, row 1 reference features)."""
    B, C, H, W = x.shape
    res = x
    h = L.group_norm(w, p + ".norm", x, cfg.norm_num_groups, 1e-6)
    h = L.conv2d(w, p + ".proj_in", h, padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    tp = p + ".transformer_blocks.0"
    # 1. self-attention  (mutual_self_attention.py:177-184)
    n = L.layer_norm(w, tp + ".norm1", h)
    h = L.attention(w, tp + ".attn1", n, n, cfg.heads) + h
    # 1.5 reference attention: K/V = bank repeated over the f frames of each batch row (:186-224)
    n = L.layer_norm(w, tp + ".norm1_5", h)
    bank_f = bank.unsqueeze(1).repeat(1, f, 1, 1).reshape(B, bank.shape[1], C)
    a = L.attention(w, tp + ".attn1_5", n, bank_f, cfg.heads)
    if w_ref != 1.0:
        a = a * w_ref
    h = a + h
    # 2. audio cross-attention (:227-244)
    n = L.layer_norm(w, tp + ".norm2", h)
    a = L.attention(w, tp + ".attn2", n, ehs, cfg.heads)
    if w_aud != 1.0:
        a = a * w_aud
    h = a + h
    # 3. GEGLU feed-forward (:247)
    h = L.feed_forward(w, tp + ".ff", L.layer_norm(w, tp + ".norm3", h)) + h
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    h = L.conv2d(w, p + ".proj_out", h, padding=0)
    return h + res


def _motion_module(w, p, x, f, cfg):
    """VanillaTemporalModule -> TemporalTransformer3DModel.forward (modules/motion_module.py:146-182),
    one TemporalTransformerBlock (:236-259) whose two VersatileAttention run over the frame axis with
    an additive sinusoidal table on the normed input (:351-388, :262-277)."""
    B, C, H, W = x.shape
    b = B // f
    tp = p + ".temporal_transformer"
    res = x
    h = L.group_norm(w, tp + ".norm", x, cfg.norm_num_groups, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = L.linear(w, tp + ".proj_in", h)
    bp = tp + ".transformer_blocks.0"
    for i in range(2):
        n = L.layer_norm(w, f"{bp}.norms.{i}", h)
        ap = f"{bp}.attention_blocks.{i}"
        # (b f) d c -> (b d) f c ; + pe[:, :f]
        t = n.reshape(b, f, H * W, C).permute(0, 2, 1, 3).reshape(b * H * W, f, C)
        t = t + w[ap + ".pos_encoder.pe"][:, :f]
        t = L.attention(w, ap, t, t, cfg.heads)
        t = t.reshape(b, H * W, f, C).permute(0, 2, 1, 3).reshape(B, H * W, C)
        h = t + h
    h = L.feed_forward(w, bp + ".ff", L.layer_norm(w, bp + ".ff_norm", h)) + h
    h = L.linear(w, tp + ".proj_out", h)
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return h + res


def unet3d_forward(w, cfg: UNetConfig, sample, timestep, encoder_hidden_states, kps_features, banks,
                   reference_attention_weight=1.0, audio_attention_weight=1.0):
    """sample [b,4,f,h,w], timestep scalar, encoder_hidden_states [b*f,5,768], kps_features [b,320,f,h,w]|None,
    banks {block_prefix: [b, hw, C]} -> [b,4,f,h,w]   (modules/unet_3d.py:400-578)."""
    b, _, f, H, W = sample.shape
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    t = torch.as_tensor(timestep).reshape(-1)[:1].expand(b)
    emb = L.time_embedding(w, t, cfg.block_out_channels[0])              # [b, 1280]
    emb_f = emb.repeat_interleave(f, dim=0)                               # one row per (b f) frame
    x = sample.permute(0, 2, 1, 3, 4).reshape(b * f, -1, H, W)           # b c f h w -> (b f) c h w
    x = L.conv2d(w, "conv_in", x)
    if kps_features is not None:
        x = x + kps_features.permute(0, 2, 1, 3, 4).reshape(b * f, -1, H, W)   # unet_3d.py:486-487
    ehs = encoder_hidden_states
    plan = block_plan(cfg)
    skips = [x]
    for blk in plan["down"]:
        p = blk["prefix"]
        for j, _ in enumerate(blk["layers"]):
            x = L.resnet(w, f"{p}.resnets.{j}", x, emb_f, g, eps)
            if blk["attn"]:
                ap = f"{p}.attentions.{j}"
                x = _spatial_transformer_read(w, ap, x, f, ehs, banks[ap], cfg,
                                              reference_attention_weight, audio_attention_weight)
            x = _motion_module(w, f"{p}.motion_modules.{j}", x, f, cfg)
            skips.append(x)
        if blk["sampler"]:
            x = L.downsample(w, f"{p}.downsamplers.0", x)
            skips.append(x)
    # mid (unet_3d_blocks.py:269-293)
    x = L.resnet(w, "mid_block.resnets.0", x, emb_f, g, eps)
    x = _spatial_transformer_read(w, "mid_block.attentions.0", x, f, ehs, banks["mid_block.attentions.0"], cfg,
                                  reference_attention_weight, audio_attention_weight)
    x = _motion_module(w, "mid_block.motion_modules.0", x, f, cfg)
    x = L.resnet(w, "mid_block.resnets.1", x, emb_f, g, eps)
    for blk in plan["up"]:
        p = blk["prefix"]
        for j, _ in enumerate(blk["layers"]):
            x = torch.cat([x, skips.pop()], dim=1)                        # unet_3d_blocks.py:694,831
            x = L.resnet(w, f"{p}.resnets.{j}", x, emb_f, g, eps)
            if blk["attn"]:
                ap = f"{p}.attentions.{j}"
                x = _spatial_transformer_read(w, ap, x, f, ehs, banks[ap], cfg,
                                              reference_attention_weight, audio_attention_weight)
            x = _motion_module(w, f"{p}.motion_modules.{j}", x, f, cfg)
        if blk["sampler"]:
            x = L.upsample(w, f"{p}.upsamplers.0", x)
    x = F.silu(L.group_norm(w, "conv_norm_out", x, g, eps))
    x = L.conv2d(w, "conv_out", x)
    return x.reshape(b, f, -1, H, W).permute(0, 2, 1, 3, 4)


# ----------------------------------------------------------------------------- 2-D ReferenceNet (bank writer)
def _spatial_transformer_write(w, p, x, ehs, cfg, banks):
    """Transformer2DModel.forward (modules/transformer_2d.py:216-399) + patched write branch
    (modules/mutual_self_attention.py:145-174 then the common FF tail :269-284)."""
    B, C, H, W = x.shape
    res = x
    h = L.group_norm(w, p + ".norm", x, cfg.norm_num_groups, 1e-6)
    h = L.conv2d(w, p + ".proj_in", h, padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    tp = p + ".transformer_blocks.0"
    n = L.layer_norm(w, tp + ".norm1", h)
    h = L.attention(w, tp + ".attn1", n, n, cfg.heads) + h
    n = L.layer_norm(w, tp + ".norm2", h)
    banks[p] = n.clone()                                                  # mutual_self_attention.py:165
    h = L.attention(w, tp + ".attn2", n, ehs, cfg.heads) + h
    h = L.feed_forward(w, tp + ".ff", L.layer_norm(w, tp + ".norm3", h)) + h
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    h = L.conv2d(w, p + ".proj_out", h, padding=0)
    return h + res


def refnet_banks(w, cfg: UNetConfig, ref_latents, timestep=0, encoder_hidden_states=None):
    """ReferenceNet forward as the pipeline calls it (pipelines/v_express_pipeline.py:502-508):
    timestep=0, text context zeros [1,1,768]; returns {block_prefix: [1, hw, C]} (16 entries)."""
    b = ref_latents.shape[0]
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    if encoder_hidden_states is None:
        encoder_hidden_states = torch.zeros(b, 1, cfg.cross_attention_dim, dtype=ref_latents.dtype)
    t = torch.as_tensor(timestep).reshape(-1)[:1].expand(b)
    emb = L.time_embedding(w, t, cfg.block_out_channels[0])
    banks = {}
    x = L.conv2d(w, "conv_in", ref_latents)
    plan = block_plan(cfg)
    skips = [x]
    for blk in plan["down"]:
        p = blk["prefix"]
        for j, _ in enumerate(blk["layers"]):
            x = L.resnet(w, f"{p}.resnets.{j}", x, emb, g, eps)
            if blk["attn"]:
                x = _spatial_transformer_write(w, f"{p}.attentions.{j}", x, encoder_hidden_states, cfg, banks)
            skips.append(x)
        if blk["sampler"]:
            x = L.downsample(w, f"{p}.downsamplers.0", x)
            skips.append(x)
    x = L.resnet(w, "mid_block.resnets.0", x, emb, g, eps)
    x = _spatial_transformer_write(w, "mid_block.attentions.0", x, encoder_hidden_states, cfg, banks)
    x = L.resnet(w, "mid_block.resnets.1", x, emb, g, eps)
    for blk in plan["up"]:
        p = blk["prefix"]
        for j, _ in enumerate(blk["layers"]):
            x = torch.cat([x, skips.pop()], dim=1)
            x = L.resnet(w, f"{p}.resnets.{j}", x, emb, g, eps)
            if blk["attn"]:
                x = _spatial_transformer_write(w, f"{p}.attentions.{j}", x, encoder_hidden_states, cfg, banks)
        if blk["sampler"]:
            x = L.upsample(w, f"{p}.upsamplers.0", x)
    # conv_norm_out is None in the ReferenceNet (unet_2d_condition.py:650) and the output is discarded.
    return banks


def reader_banks(banks, do_classifier_free_guidance=True):
    """ReferenceAttentionControl.update (modules/mutual_self_attention.py:357-363):
    CFG -> cat([zeros_like(v), v]) so batch row 0 (uncond) sees an all-zero bank."""
    if do_classifier_free_guidance:
        return {k: torch.cat([torch.zeros_like(v), v]) for k, v in banks.items()}
    return {k: v.clone() for k, v in banks.items()}
