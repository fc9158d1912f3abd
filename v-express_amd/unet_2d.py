"""ReferenceNet: drop-in for the reference `UNet2DConditionModel` (modules/unet_2d_condition.py:877-1313) as the
pipeline uses it — one forward per clip at timestep 0 whose only product is the 16 reference-feature banks
written by the patched BasicTransformerBlocks (modules/mutual_self_attention.py:145-174).

V-Express deltas kept: `conv_norm_out` is None (unet_2d_condition.py:650) and the sample output is discarded
(pipelines/v_express_pipeline.py:503-508), so the tail after the last transformer block is not computed.
"""
import json

import torch

from . import blocks as B
from . import ops
from .synth import UNetConfig, block_plan
from .unet_3d import _UNetBase, _config_from_dict


class UNet2DConditionModel(_UNetBase):
    THREE_D = False

    def __init__(self, cfg: UNetConfig = None, config_dict=None):
        super().__init__(cfg or UNetConfig(), config_dict)

    @classmethod
    def from_config(cls, config, **kwargs):
        if isinstance(config, dict):
            cd = dict(config)
        else:
            with open(config) as f:
                cd = json.load(f)
        cd.update(kwargs)
        return cls(_config_from_dict(cd), cd)

    def forward(self, sample, timestep, encoder_hidden_states, return_dict=True, **unused):
        """sample [n,4,h,w]; in *write* mode fills `self.banks[block_prefix] = [n*hw, C]` (bf16).  The UNet output
        itself is never used by V-Express; a zero tensor of the right shape is returned for API compatibility."""
        P, cfg = self._prepared(), self.cfg
        if self.reference_mode != "write":
            raise RuntimeError("ReferenceNet is only built as the bank writer: wrap it in "
                               "ReferenceAttentionControl(mode='write') first")
        g, eps, heads = cfg.norm_num_groups, cfg.norm_eps, cfg.heads
        n, c, H, W = sample.shape
        if H % 8 or W % 8:
            raise ValueError("latent height/width must be multiples of 8")
        dev = self._device
        x_in = ops.ncfhw_to_nhwc(sample.to(dev).float().unsqueeze(2), 8)          # [n, hw, 8]
        ehs = encoder_hidden_states.to(device=dev, dtype=ops.BF16)
        ehs = ehs.reshape(-1, ehs.shape[-1]).contiguous()                          # [n*n_ctx, 768]
        rows = self.time_rows(timestep, n)
        frames, h_, w_ = n, H, W
        x = ops.gemm(x_in.view(frames * H * W, -1), P.conv_in.w, P.conv_in.b,
                     geom=ops.ConvGeom(frames, H, W, 3, 3, 1, 1)).view(frames, H * W, -1)
        plan = block_plan(cfg)
        banks = {}
        skips = [(x, h_, w_)]
        # the last transformer block of the last up block is the final bank; nothing after it is needed
        last_attn = None
        for blk in plan["up"]:
            if blk["attn"]:
                last_attn = f"{blk['prefix']}.attentions.{len(blk['layers']) - 1}"

        def layer(p, j, attn, x, skip):
            x = B.resnet_block(P[f"{p}.resnets.{j}"], x, frames, h_, w_, groups=g, eps=eps,
                               temb=self._temb(rows, f"{p}.resnets.{j}"), rows_per_group=h_ * w_, skip=skip)
            if attn:
                ap = f"{p}.attentions.{j}"
                x, bank = B.spatial_transformer_write(P[ap], x, frames=frames, H=h_, W=w_, heads=heads, groups=g,
                                                      ehs=ehs)
                banks[ap] = bank
            return x

        for blk in plan["down"]:
            p = blk["prefix"]
            for j, _ in enumerate(blk["layers"]):
                x = layer(p, j, blk["attn"], x, None)
                skips.append((x, h_, w_))
            if blk["sampler"]:
                x, h_, w_ = B.downsample(P[f"{p}.downsamplers.0"], x, frames, h_, w_)
                skips.append((x, h_, w_))
        x = B.resnet_block(P["mid_block.resnets.0"], x, frames, h_, w_, groups=g, eps=eps,
                           temb=self._temb(rows, "mid_block.resnets.0"), rows_per_group=h_ * w_)
        x, bank = B.spatial_transformer_write(P["mid_block.attentions.0"], x, frames=frames, H=h_, W=w_, heads=heads,
                                              groups=g, ehs=ehs)
        banks["mid_block.attentions.0"] = bank
        x = B.resnet_block(P["mid_block.resnets.1"], x, frames, h_, w_, groups=g, eps=eps,
                           temb=self._temb(rows, "mid_block.resnets.1"), rows_per_group=h_ * w_)
        done = False
        for blk in plan["up"]:
            p = blk["prefix"]
            for j, _ in enumerate(blk["layers"]):
                skip, sh, sw = skips.pop()
                x = layer(p, j, blk["attn"], x, skip)
                if f"{p}.attentions.{j}" == last_attn:
                    done = True
                    break
            if done:
                break
            if blk["sampler"]:
                x, h_, w_ = B.upsample(P[f"{p}.upsamplers.0"], x, frames, h_, w_, items=frames)
        self.banks = banks
        out = torch.zeros_like(sample)
        return (out,) if not return_dict else type("UNet2DConditionOutput", (), {"sample": out})()

    __call__ = forward
