"""sd-vae-ft-mse decoder (diffusers AutoencoderKL.decode as called by VExpressPipeline.decode_latents,
pipelines/v_express_pipeline.py:152-166) on libvexpress_hip kernels.

The arithmetic lives in diffusers==0.29.2 (absent third-party dependency):
post_quant_conv 1x1 -> conv_in 3x3 -> mid (resnet, single-head attention d=512 with GroupNorm pre-norm, biased
q/k/v/out and a residual, resnet) -> 4 up blocks x 3 resnets (+ nearest-2x conv3x3 after the first three) ->
GroupNorm(eps 1e-6) -> SiLU -> conv_out.  The reference decodes one frame at a time and copies each to the
host (:158-162); here frames are decoded in batches on the device and returned as [n, 3, H, W] fp32.
"""
from types import SimpleNamespace

import torch

from . import blocks as B
from . import ops
from . import weights as Wt
from .module_base import DeviceModule
from .synth import VaeConfig


class AutoencoderKLDecoder(DeviceModule):
    """Decoder half of AutoencoderKL with the reference's `vae.decode(z).sample` / `vae.config` surface."""

    def __init__(self, cfg: VaeConfig = None):
        super().__init__()
        self.cfg = cfg or VaeConfig()
        self.config = SimpleNamespace(block_out_channels=tuple(self.cfg.block_out_channels),
                                      scaling_factor=self.cfg.scaling_factor,
                                      latent_channels=self.cfg.latent_channels)
        self._PE = None

    def _invalidate(self):
        self._P = None
        self._PE = None

    _PREFIXES = ("decoder.", "post_quant_conv.")

    def load_state_dict(self, sd, strict=False):
        keep = {k: v.detach() for k, v in sd.items() if k.startswith(self._PREFIXES)}
        self._raw.update(keep)
        self._released = False
        self._invalidate()
        return SimpleNamespace(missing_keys=[], unexpected_keys=[k for k in sd if k not in keep])

    def init_random(self, seed=44):
        from . import synth
        self.load_state_dict(synth.vae_decoder_state_dict(self.cfg, seed=seed))
        return self

    def _prepared(self):
        if self._P is not None:
            return self._P
        self._need_gpu()
        sd, dev, cfg = self._raw, self._device, self.cfg
        P = Wt.Prepared()
        P["post_quant"] = Wt.prep_conv(sd, "post_quant_conv", dev)
        P["conv_in"] = Wt.prep_conv(sd, "decoder.conv_in", dev)
        for j in range(2):
            P[f"mid.resnets.{j}"] = Wt.prep_resnet(sd, f"decoder.mid_block.resnets.{j}", dev)
        a = "decoder.mid_block.attentions.0"
        P["mid.attn"] = Wt.Prepared(norm=Wt.prep_norm(sd, a + ".group_norm", dev), attn=Wt.prep_self_attn(sd, a, dev))
        n = len(cfg.block_out_channels)
        for i in range(n):
            for j in range(cfg.layers_per_block + 1):
                P[f"up.{i}.resnets.{j}"] = Wt.prep_resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", dev)
            if i != n - 1:
                P[f"up.{i}.upsampler"] = Wt.prep_conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", dev)
                P[f"up.{i}.upsampler"]["phases"] = Wt.fold_upsample_phases(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", dev)
        P["norm_out"] = Wt.prep_norm(sd, "decoder.conv_norm_out", dev)
        P["conv_out"] = Wt.prep_conv(sd, "decoder.conv_out", dev)
        self._P = P
        return P

    def decode_tokens(self, z_tokens, n, H, W):
        """z_tokens: bf16 [n, hw, 8] (latent channels zero-padded; already divided by the scaling factor)
        -> fp32 [n*(8H)*(8W), 8] NHWC (columns >= 3 are zero)."""
        P, cfg = self._prepared(), self.cfg
        g = cfg.norm_num_groups
        hw = H * W
        x = ops.gemm(z_tokens.view(n * hw, -1), P.post_quant.w, P.post_quant.b)
        x = ops.gemm(x, P.conv_in.w, P.conv_in.b, geom=ops.ConvGeom(n, H, W, 3, 3, 1, 1)).view(n, hw, -1)
        x = B.resnet_block(P["mid.resnets.0"], x, n, H, W, groups=g, eps=1e-6)
        # single-head attention with GroupNorm pre-norm and residual
        c = x.shape[-1]
        A = P["mid.attn"]
        nrm = ops.groupnorm(x, A.norm.g, A.norm.b, frames=n, hw=hw, groups=g, eps=1e-6, silu=False)
        h = x.reshape(n * hw, c).clone()
        B._self_attention(A.attn, nrm.view(n * hw, c), h, seqs=n, n_tok=hw, heads=1)
        x = h.view(n, hw, c)
        x = B.resnet_block(P["mid.resnets.1"], x, n, H, W, groups=g, eps=1e-6)
        nb = len(cfg.block_out_channels)
        for i in range(nb):
            for j in range(cfg.layers_per_block + 1):
                x = B.resnet_block(P[f"up.{i}.resnets.{j}"], x, n, H, W, groups=g, eps=1e-6)
            if i != nb - 1:
                x, H, W = B.upsample(P[f"up.{i}.upsampler"], x, n, H, W, items=n)
        hw = H * W
        nrm = ops.groupnorm(x, P.norm_out.g, P.norm_out.b, frames=n, hw=hw, groups=g, eps=1e-6, silu=True,
                            pad_hw=(H, W))
        out = ops.gemm(nrm.view(n * (H + 2) * (W + 2), -1), P.conv_out.w, P.conv_out.b,
                       geom=ops.ConvGeom(n, H + 2, W + 2, 3, 3, 1, 0), out_f32=True)
        return out, H, W

    def decode(self, z):
        """z: [n, 4, h, w] (any float dtype/device) -> SimpleNamespace(sample=[n, 3, 8h, 8w] fp32 on device)."""
        n, c, h, w = z.shape
        zt = ops.ncfhw_to_nhwc(z.to(self._device).float().unsqueeze(2), 8)
        out, H, W = self.decode_tokens(zt, n, h, w)
        img = ops.nhwc_to_ncfhw(out, n, self.cfg.out_channels, 1, H, W)[:, :, 0]
        return SimpleNamespace(sample=img)

    def decode_video(self, latents, chunk=8):
        """VExpressPipeline.decode_latents (pipelines/v_express_pipeline.py:152-166) on device:
        latents fp32 [1, 4, F, h, w] -> video fp32 [1, 3, F, 8h, 8w] in [0, 1]."""
        b, c, F, h, w = latents.shape
        lat = (latents.float() / self.cfg.scaling_factor)
        frames = []
        for f0 in range(0, F, chunk):
            part = lat[:, :, f0:f0 + chunk]
            n = part.shape[2] * b
            zt = ops.ncfhw_to_nhwc(part.contiguous(), 8)
            out, H, W = self.decode_tokens(zt, n, h, w)
            frames.append(ops.vae_postprocess(out, n, self.cfg.out_channels, H, W))
        video = torch.cat(frames)                                    # [(b F), 3, H, W]
        return video.view(b, F, self.cfg.out_channels, video.shape[-2], video.shape[-1]).permute(0, 2, 1, 3, 4)


class AutoencoderKL(AutoencoderKLDecoder):
    """Decoder + ENCODER halves: `vae.encode(x).latent_dist.mean` as called by
    VExpressPipeline.prepare_reference_latent (pipelines/v_express_pipeline.py:343-348).  diffusers AutoencoderKL
    encoder, restated: conv_in -> 4 down blocks of 2 resnets (Downsample2D(padding=0): F.pad (0,1,0,1) + conv3x3
    stride 2 after the first three) -> mid (resnet, single-head attention, resnet) -> GroupNorm(eps 1e-6) -> SiLU ->
    conv_out (2 x latent channels) -> quant_conv 1x1; the posterior mean is the first half of the channels."""
    _PREFIXES = ("decoder.", "post_quant_conv.", "encoder.", "quant_conv.")

    def _prepared_encoder(self):
        if self._PE is not None:
            return self._PE
        self._need_gpu()
        sd, dev, cfg = self._raw, self._device, self.cfg
        P = Wt.Prepared()
        P["conv_in"] = Wt.prep_conv(sd, "encoder.conv_in", dev)
        n = len(cfg.block_out_channels)
        for i in range(n):
            for j in range(cfg.layers_per_block):
                P[f"down.{i}.resnets.{j}"] = Wt.prep_resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", dev)
            if i != n - 1:
                P[f"down.{i}.downsampler"] = Wt.prep_conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", dev)
        for j in range(2):
            P[f"mid.resnets.{j}"] = Wt.prep_resnet(sd, f"encoder.mid_block.resnets.{j}", dev)
        a = "encoder.mid_block.attentions.0"
        P["mid.attn"] = Wt.Prepared(norm=Wt.prep_norm(sd, a + ".group_norm", dev), attn=Wt.prep_self_attn(sd, a, dev))
        P["norm_out"] = Wt.prep_norm(sd, "encoder.conv_norm_out", dev)
        P["conv_out"] = Wt.prep_conv(sd, "encoder.conv_out", dev)
        P["quant"] = Wt.prep_conv(sd, "quant_conv", dev)
        self._PE = P
        return P

    def encode(self, x):
        """x: [n, 3, H, W] in [-1, 1] -> SimpleNamespace(latent_dist=SimpleNamespace(mean=[n, 4, H/8, W/8] fp32))."""
        P, cfg = self._prepared_encoder(), self.cfg
        g = cfg.norm_num_groups
        n, c, H, W = x.shape
        t = ops.ncfhw_to_nhwc(x.to(self._device).float().unsqueeze(2).contiguous(), 8)
        t = ops.gemm(t.view(n * H * W, 8), P.conv_in.w, P.conv_in.b, geom=ops.ConvGeom(n, H, W, 3, 3, 1, 1))
        t = t.view(n, H * W, -1)
        nb = len(cfg.block_out_channels)
        for i in range(nb):
            for j in range(cfg.layers_per_block):
                t = B.resnet_block(P[f"down.{i}.resnets.{j}"], t, n, H, W, groups=g, eps=1e-6)
            if i != nb - 1:
                d = P[f"down.{i}.downsampler"]
                geom = ops.ConvGeom(n, H, W, 3, 3, 2, 0, pad_end=1)
                t = ops.gemm(t.view(n * H * W, -1), d.w, d.b, geom=geom)
                H, W = geom.h_out, geom.w_out
                t = t.view(n, H * W, -1)
        hw = H * W
        t = B.resnet_block(P["mid.resnets.0"], t, n, H, W, groups=g, eps=1e-6)
        cch = t.shape[-1]
        A = P["mid.attn"]
        nrm = ops.groupnorm(t, A.norm.g, A.norm.b, frames=n, hw=hw, groups=g, eps=1e-6, silu=False)
        h = t.reshape(n * hw, cch).clone()
        B._self_attention(A.attn, nrm.view(n * hw, cch), h, seqs=n, n_tok=hw, heads=1)
        t = B.resnet_block(P["mid.resnets.1"], h.view(n, hw, cch), n, H, W, groups=g, eps=1e-6)
        nrm = ops.groupnorm(t, P.norm_out.g, P.norm_out.b, frames=n, hw=hw, groups=g, eps=1e-6, silu=True,
                            pad_hw=(H, W))
        m = ops.gemm(nrm.view(n * (H + 2) * (W + 2), -1), P.conv_out.w, P.conv_out.b,
                     geom=ops.ConvGeom(n, H + 2, W + 2, 3, 3, 1, 0))
        m = ops.gemm(m, P.quant.w, P.quant.b, out_f32=True)                     # [n*hw, 8] fp32: mean | logvar
        mean = ops.nhwc_to_ncfhw(m, n, cfg.latent_channels, 1, H, W)[:, :, 0]
        return SimpleNamespace(latent_dist=SimpleNamespace(mean=mean))
