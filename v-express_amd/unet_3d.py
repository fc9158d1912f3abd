"""Drop-in for the reference `UNet3DConditionModel` (modules/unet_3d.py) on libvexpress_hip kernels.

Same construction / call surface as the reference (surface #2):
`UNet3DConditionModel.from_config_2d(unet_config, unet_additional_kwargs)`, `.to(dtype, device)`,
`.load_state_dict(sd, strict=False)` (reference key names, may be called repeatedly: SD-1.5 weights, then
V-Express denoising weights, then motion-module weights — inference.py:86-93), `.config.cross_attention_dim`,
`.in_channels`, `.dtype`, `.device`, and

    forward(sample[b,4,f,h,w], timestep, encoder_hidden_states[b*f,5,768], class_labels=None,
            kps_features[b,320,f,h,w]=None, ..., return_dict=True) -> UNet3DConditionOutput(sample) | (sample,)

The reference's `ReferenceAttentionControl` monkey-patches diffusers blocks; here the *read* branch
(modules/mutual_self_attention.py:176-267) is the native block forward and `ReferenceAttentionControl`
(mutual_self_attention.py in this package) only installs banks and the two attention weights.
"""
import json
import math
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from . import blocks as B
from . import lib as L
from . import ops
from . import weights as Wt
from .module_base import DeviceModule
from .synth import UNetConfig, block_plan


@dataclass
class UNet3DConditionOutput:
    sample: torch.Tensor


def _config_from_dict(d):
    d = dict(d)
    ahd = d.get("attention_head_dim", 8)
    if not isinstance(ahd, int):
        if len(set(ahd)) != 1:
            raise ValueError("per-level attention_head_dim is not supported")
        ahd = ahd[0]
    down = d.get("down_block_types")
    attn_levels = tuple("CrossAttn" in t for t in down) if down else (True, True, True, False)
    return UNetConfig(in_channels=d.get("in_channels", 4), out_channels=d.get("out_channels", 4),
                      block_out_channels=tuple(d.get("block_out_channels", (320, 640, 1280, 1280))),
                      layers_per_block=d.get("layers_per_block", 2), heads=ahd,
                      cross_attention_dim=d.get("cross_attention_dim", 768),
                      norm_num_groups=d.get("norm_num_groups", 32), norm_eps=d.get("norm_eps", 1e-5),
                      attn_levels=attn_levels)


class _UNetBase(DeviceModule):
    """State shared by the denoising UNet and the ReferenceNet: config, raw/prepared weights, time embedding."""

    THREE_D = True

    def __init__(self, cfg: UNetConfig, config_dict=None):
        super().__init__()
        self.cfg = cfg
        cd = dict(config_dict or {})
        cd.setdefault("cross_attention_dim", cfg.cross_attention_dim)
        cd.setdefault("in_channels", cfg.in_channels)
        cd.setdefault("center_input_sample", False)
        cd.setdefault("class_embed_type", None)
        self.config = SimpleNamespace(**cd)
        self.in_channels = cfg.in_channels
        self._expected = None
        self._temb_cache = {}
        # installed by ReferenceAttentionControl
        self.reference_mode = None
        self.reference_attention_weight = 1.0
        self.audio_attention_weight = 1.0
        self.banks = {}

    def expected_keys(self):
        if self._expected is None:
            from . import synth
            sd = (synth.unet3d_state_dict if self.THREE_D else synth.refnet_state_dict)(self.cfg, device="meta")
            self._expected = {k: tuple(v.shape) for k, v in sd.items()}
        return self._expected

    def _invalidate(self):
        self._P = None
        self._temb_cache.clear()

    def init_random(self, seed=42):
        """Random-init weights with the reference schema (no checkpoints exist offline)."""
        from . import synth
        gen = synth.unet3d_state_dict if self.THREE_D else synth.refnet_state_dict
        self.load_state_dict(gen(self.cfg, seed=seed), strict=True)
        return self

    # ---- weight preparation
    def _prepared(self):
        if self._P is not None:
            return self._P
        self._need_gpu()
        missing = [k for k in self.expected_keys() if k not in self._raw]
        if missing:
            raise RuntimeError(f"weights not loaded (or released): {len(missing)} tensors missing, e.g. {missing[:3]}")
        sd, dev, cfg = self._raw, self._device, self.cfg
        P = Wt.Prepared()
        P["conv_in"] = Wt.prep_conv(sd, "conv_in", dev)
        P["time1"] = Wt.prep_linear(sd, "time_embedding.linear_1", dev)
        P["time2"] = Wt.prep_linear(sd, "time_embedding.linear_2", dev)
        plan = block_plan(cfg)
        temb_w, temb_b, off = [], [], 0
        P["temb_off"] = {}

        def add_resnet(p):
            nonlocal off
            P[p] = Wt.prep_resnet(sd, p, dev)
            w = sd[p + ".time_emb_proj.weight"]
            temb_w.append(w.detach().to(device=dev, dtype=Wt.BF16))
            temb_b.append(sd[p + ".time_emb_proj.bias"].detach().to(device=dev, dtype=torch.float32))
            P["temb_off"][p] = (off, w.shape[0])
            off += w.shape[0]

        spatial = Wt.prep_spatial_read if self.THREE_D else Wt.prep_spatial_write
        for blk in plan["down"] + plan["up"]:
            p = blk["prefix"]
            for j, _ in enumerate(blk["layers"]):
                add_resnet(f"{p}.resnets.{j}")
                if blk["attn"]:
                    P[f"{p}.attentions.{j}"] = spatial(sd, f"{p}.attentions.{j}", dev, cfg.heads)
                if self.THREE_D:
                    P[f"{p}.motion_modules.{j}"] = Wt.prep_motion(sd, f"{p}.motion_modules.{j}", dev)
            if blk["sampler"]:
                name = "downsamplers" if p.startswith("down") else "upsamplers"
                P[f"{p}.{name}.0"] = Wt.prep_conv(sd, f"{p}.{name}.0.conv", dev)
                if name == "upsamplers":
                    P[f"{p}.{name}.0"]["phases"] = Wt.fold_upsample_phases(sd, f"{p}.{name}.0.conv", dev)
        add_resnet("mid_block.resnets.0")
        P["mid_block.attentions.0"] = spatial(sd, "mid_block.attentions.0", dev, cfg.heads)
        if self.THREE_D:
            P["mid_block.motion_modules.0"] = Wt.prep_motion(sd, "mid_block.motion_modules.0", dev)
        add_resnet("mid_block.resnets.1")
        P["temb_w"] = torch.cat(temb_w, dim=0).contiguous()
        P["temb_b"] = torch.cat(temb_b, dim=0).contiguous()
        if self.THREE_D:
            P["conv_norm_out"] = Wt.prep_norm(sd, "conv_norm_out", dev)
            P["conv_out"] = Wt.prep_conv(sd, "conv_out", dev)
        self._P = P
        self._temb_cache.clear()
        return P

    # ---- time embedding: Timesteps + TimestepEmbedding (modules/unet_3d.py:464-470) and all 22
    #      ResnetBlock3D.time_emb_proj(silu(emb)) rows (modules/resnet.py:225-233), once per timestep value
    def time_rows(self, timestep, batch):
        P = self._prepared()
        t = float(timestep.item()) if torch.is_tensor(timestep) else float(timestep)
        key = (t, batch)
        if key in self._temb_cache:
            return self._temb_cache[key]
        c0 = self.cfg.block_out_channels[0]
        half = c0 // 2
        expo = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        ang = torch.full((batch, 1), t, dtype=torch.float32) * expo[None]
        emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)      # flip_sin_to_cos=True
        emb = emb.to(device=self._device, dtype=ops.BF16)
        e1 = ops.gemm(emb, P.time1.w, P.time1.b, act=L.VX_ACT_SILU)
        e2 = ops.gemm(e1, P.time2.w, P.time2.b, act=L.VX_ACT_SILU)      # every consumer applies SiLU first
        rows = ops.gemm(e2, P.temb_w, P.temb_b, out_f32=True)           # fp32 [batch, sum Cout]
        if len(self._temb_cache) > 64:
            self._temb_cache.clear()
        self._temb_cache[key] = rows
        return rows

    def _temb(self, rows, p):
        off, n = self._P["temb_off"][p]
        return rows[:, off:off + n]


class UNet3DConditionModel(_UNetBase):
    THREE_D = True

    def __init__(self, cfg: UNetConfig = None, config_dict=None, **unet_additional_kwargs):
        super().__init__(cfg or UNetConfig(), config_dict)
        mm = (unet_additional_kwargs or {}).get("motion_module_kwargs", {}) or {}
        self.cfg.temporal_max_len = mm.get("temporal_position_encoding_max_len", self.cfg.temporal_max_len)
        if unet_additional_kwargs:
            if not unet_additional_kwargs.get("use_motion_module", True):
                raise NotImplementedError("only the inference_v2.yaml architecture (motion modules on) is built")
            if unet_additional_kwargs.get("unet_use_temporal_attention", False):
                raise NotImplementedError("unet_use_temporal_attention=True is dead in the reference config")
        self._kps_cache = None
        self.fp8_projections = False

    @classmethod
    def from_config_2d(cls, unet_config_path, unet_additional_kwargs=None):
        """modules/unet_3d.py:673-698: SD-1.5 2-D UNet config + inference_v2.yaml `unet_additional_kwargs`."""
        if isinstance(unet_config_path, dict):
            cd = dict(unet_config_path)
        else:
            with open(unet_config_path) as f:
                cd = json.load(f)
        cd["down_block_types"] = ["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"]
        cd["up_block_types"] = ["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3
        cd["mid_block_type"] = "UNetMidBlock3DCrossAttn"
        return cls(_config_from_dict(cd), cd, **(unet_additional_kwargs or {}))

    from_config = from_config_2d

    # ---- hot path on resident token tensors (used by the pipeline; no layout changes inside)
    def precompute_audio_kv(self, ehs):
        """Audio cross-attention K | V of every spatial transformer block for one batch of audio tokens
        (`ehs` bf16 [b*f*n_ctx, 768]) -> {block prefix: [b*f*n_ctx, 2C]}.  Step-invariant (like the reference-attention
        banks): the loop passes the result to `forward_tokens(audio_kv=...)` for all DDIM steps of a window."""
        P = self._prepared()
        plan = block_plan(self.cfg)
        out = {}
        for blk in plan["down"] + plan["up"]:
            if blk["attn"]:
                for j, _ in enumerate(blk["layers"]):
                    ap = f"{blk['prefix']}.attentions.{j}"
                    out[ap] = B.audio_kv(P[ap], ehs)
        out["mid_block.attentions.0"] = B.audio_kv(P["mid_block.attentions.0"], ehs)
        return out

    def forward_tokens(self, *args, **kwargs):
        """`_forward_tokens` under the model's fp8 switch: with `self.fp8_projections = True` the q / k / v / out
        projections of every attention (attn1, attn1_5, attn2, both temporal attentions) run on the fp8 MFMA GEMM with
        per-row e4m3 operands (BASELINE.json configs[4]); everything else stays bf16."""
        with ops.fp8_projections(self.fp8_projections):
            return self._forward_tokens(*args, **kwargs)

    def _forward_tokens(self, x_in, timestep, ehs, kps, *, b, f, H, W, batch_rows=None, audio_kv=None,
                        audio_zero=None, frame_shard=None):
        """x_in: bf16 [b*f, HW, 8] (latent channels zero-padded), ehs: bf16 [b*f*n_ctx, 768],
        kps: bf16 [b*f, HW, C0] or None -> fp32 [b*f*HW, 8] (columns >= out_channels are zero).
        batch_rows: which rows of the installed banks the b batch rows use (default 0..b-1; a lone CFG half
        running on another GPU passes [0] or [1]).  audio_kv: `precompute_audio_kv(ehs)`; audio_zero: per batch row,
        True when that row's audio tokens are all zero (its audio cross-attention then reduces to the output bias).
        frame_shard (distributed.FrameShard): all tensors hold only this rank's f of the window's f * size frames; the
        motion modules exchange layouts inside the shard group."""
        P, cfg = self._prepared(), self.cfg
        g, eps, heads = cfg.norm_num_groups, cfg.norm_eps, cfg.heads
        frames = b * f
        f_window = f * (frame_shard.size if frame_shard is not None else 1)
        if f_window > cfg.temporal_max_len:
            raise ValueError(f"window length {f_window} exceeds the positional-encoding table "
                             f"({cfg.temporal_max_len})")
        if frame_shard is not None and ((H // 8) * (W // 8)) % frame_shard.size:
            raise ValueError(f"{frame_shard.size} frame shards do not divide the {H // 8}x{W // 8} tokens of the "
                             "coarsest level")
        rows = self.time_rows(timestep, b)
        rpg_scale = f
        banks = self.banks
        if self.reference_mode != "read":
            raise RuntimeError("denoising UNet needs ReferenceAttentionControl(mode='read').update(writer) first")
        w_ref, w_aud = self.reference_attention_weight, self.audio_attention_weight
        rowsel = list(batch_rows) if batch_rows is not None else list(range(b))
        if len(rowsel) != b:
            raise ValueError("batch_rows must name one bank row per batch row")
        hw = H * W
        x = ops.gemm(x_in.view(frames * hw, -1), P.conv_in.w, P.conv_in.b, geom=ops.ConvGeom(frames, H, W, 3, 3, 1, 1),
                     residual=None if kps is None else kps.view(frames * hw, -1))      # unet_3d.py:485-487
        x = x.view(frames, hw, -1)
        plan = block_plan(cfg)
        skips = [(x, H, W)]
        h_, w_ = H, W

        def layer(p, j, attn, x, skip, gn_next=False):
            x = B.resnet_block(P[f"{p}.resnets.{j}"], x, frames, h_, w_, groups=g, eps=eps,
                               temb=self._temb(rows, f"{p}.resnets.{j}"), rows_per_group=rpg_scale * h_ * w_,
                               skip=skip, items=b)
            if attn:
                ap = f"{p}.attentions.{j}"
                x = B.spatial_transformer_read(P[ap], x, b=b, f=f, H=h_, W=w_, heads=heads, groups=g, ehs=ehs,
                                               bank=[banks[ap][r] for r in rowsel], w_ref=w_ref, w_aud=w_aud,
                                               kv=None if audio_kv is None else audio_kv[ap], audio_zero=audio_zero)
            return B.motion_module(P[f"{p}.motion_modules.{j}"], x, b=b, f=f, H=h_, W=w_, heads=heads, groups=g,
                                   shard=frame_shard, gn_next=gn_next)

        # gn_next / gn_groups: the tensor is read next by a GroupNorm over exactly its own channels (a resnet's norm1
        # without a skip concat, conv_norm_out) - its producer then leaves that GroupNorm's partial sums on it
        for blk in plan["down"]:
            p = blk["prefix"]
            last = len(blk["layers"]) - 1
            for j, _ in enumerate(blk["layers"]):
                x = layer(p, j, blk["attn"], x, None, gn_next=(j < last or not blk["sampler"]))
                skips.append((x, h_, w_))
            if blk["sampler"]:
                x, h_, w_ = B.downsample(P[f"{p}.downsamplers.0"], x, frames, h_, w_, gn_groups=g)
                skips.append((x, h_, w_))
        # mid (unet_3d_blocks.py:269-293)
        x = B.resnet_block(P["mid_block.resnets.0"], x, frames, h_, w_, groups=g, eps=eps,
                           temb=self._temb(rows, "mid_block.resnets.0"), rows_per_group=rpg_scale * h_ * w_, items=b)
        x = B.spatial_transformer_read(P["mid_block.attentions.0"], x, b=b, f=f, H=h_, W=w_, heads=heads, groups=g,
                                       ehs=ehs, bank=[banks["mid_block.attentions.0"][r] for r in rowsel],
                                       w_ref=w_ref, w_aud=w_aud,
                                       kv=None if audio_kv is None else audio_kv["mid_block.attentions.0"],
                                       audio_zero=audio_zero)
        x = B.motion_module(P["mid_block.motion_modules.0"], x, b=b, f=f, H=h_, W=w_, heads=heads, groups=g,
                            shard=frame_shard, gn_next=True)
        x = B.resnet_block(P["mid_block.resnets.1"], x, frames, h_, w_, groups=g, eps=eps,
                           temb=self._temb(rows, "mid_block.resnets.1"), rows_per_group=rpg_scale * h_ * w_, items=b)
        for bi_, blk in enumerate(plan["up"]):
            p = blk["prefix"]
            for j, _ in enumerate(blk["layers"]):
                skip, sh, sw = skips.pop()
                if (sh, sw) != (h_, w_):
                    raise RuntimeError("skip resolution mismatch (sample size must be a multiple of 8 latents)")
                # (up-block resnets normalise the concat [x, skip]: only the very last output meets a plain GroupNorm)
                x = layer(p, j, blk["attn"], x, skip,
                          gn_next=(bi_ == len(plan["up"]) - 1 and j == len(blk["layers"]) - 1 and not blk["sampler"]))
            if blk["sampler"]:
                x, h_, w_ = B.upsample(P[f"{p}.upsamplers.0"], x, frames, h_, w_, items=b)
        n = ops.groupnorm(x, P.conv_norm_out.g, P.conv_norm_out.b, frames=frames, hw=hw, groups=g, eps=eps, silu=True,
                          pad_hw=(H, W))
        return ops.gemm(n.view(frames * (H + 2) * (W + 2), -1), P.conv_out.w, P.conv_out.b,
                        geom=ops.ConvGeom(frames, H + 2, W + 2, 3, 3, 1, 0), out_f32=True)

    # ---- reference call surface
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, kps_features=None,
                attention_mask=None, down_block_additional_residuals=None, mid_block_additional_residual=None,
                return_dict=True):
        if attention_mask is not None or down_block_additional_residuals is not None \
                or mid_block_additional_residual is not None or class_labels is not None:
            raise NotImplementedError("attention_mask / additional residuals / class_labels are unused by V-Express")
        b, c, f, H, W = sample.shape
        if H % 8 or W % 8:
            raise ValueError("latent height/width must be multiples of 8 (three 2x resampling stages)")
        dev = self._device
        x_in = ops.ncfhw_to_nhwc(sample.to(dev), 8)
        ehs = encoder_hidden_states.to(device=dev, dtype=ops.BF16)
        if ehs.shape[0] != b * f:
            ehs = ehs.repeat_interleave(f, dim=0)                     # transformer_3d.py:116-119
        ehs = ehs.reshape(-1, ehs.shape[-1]).contiguous()
        kps = None
        if kps_features is not None:
            kps = ops.ncfhw_to_nhwc(kps_features.to(dev), self.cfg.block_out_channels[0])
        out = self.forward_tokens(x_in, timestep, ehs, kps, b=b, f=f, H=H, W=W)
        y = ops.nhwc_to_ncfhw(out, b, self.cfg.out_channels, f, H, W).to(sample.dtype)
        if not return_dict:
            return (y,)
        return UNet3DConditionOutput(sample=y)

    __call__ = forward
