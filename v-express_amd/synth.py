"""Seeded synthetic weights and inputs with the reference's state_dict key schema.

No checkpoints exist offline, so parity tests and the benchmark use random-init
weights.  Keys and shapes follow the reference modules exactly (SURVEY.md Appendix C;
modules/unet_3d.py, unet_3d_blocks.py, attention.py:321-376, motion_module.py:119-144,207-234,
transformer_3d.py:58-95, resnet.py:157-215, unet_2d_condition.py / unet_2d_blocks.py) so a real
checkpoint loads through the same `load_state_dict` path.  tests/test_oracle_vs_reference.py checks the
schema against the reference's own `state_dict()`.

Draws: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's default Linear/Conv init; norm affine
parameters are perturbed (gamma ~ 1 + 0.1 N, beta ~ 0.1 N) and the tensors the reference
zero-initialises (attn2.to_out, motion proj_out — attention.py:361, motion_module.py:72-75) are
re-drawn N(0, 0.02^2) so that every path is numerically live.
"""
import math
from dataclasses import dataclass
from typing import Tuple

import torch


@dataclass
class UNetConfig:
    """Architecture of both UNets (SD-1.5 config + inference_v2.yaml:1-23)."""
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    heads: int = 8
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    temporal_max_len: int = 32
    attn_levels: Tuple[bool, ...] = (True, True, True, False)

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4


@dataclass
class VaeConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


class _Gen:
    def __init__(self, seed, device="cpu", dtype=torch.float32):
        self.meta = str(device) == "meta"          # shapes only (schema queries): nothing is drawn
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(seed)
        self.device = device
        self.dtype = dtype
        self.sd = {}
        # draw_on_device: use the device's own generator (fast; values differ from the CPU draw, so only for
        # benchmarks, never for parity tests that need identical weights on both sides)
        self.dg = None
        # timing_only (bench.py's CPU leg): values come out of one pre-drawn pool at rotating offsets - the same
        # distributions, every tensor its own memory, tensors repeat each other's values; ~5x faster than 5.4 GB of
        # sequential CPU draws.  Never for parity tests.
        self.pool_u = self.pool_n = None
        self.pool_at = 0

    def use_pool(self, numel=1 << 25):
        self.pool_u = torch.rand(numel, generator=self.g) * 2 - 1
        self.pool_n = torch.randn(numel, generator=self.g)

    def _from_pool(self, pool, shape):
        n = 1
        for d in shape:
            n *= d
        if n > pool.numel():
            return pool.repeat((n + pool.numel() - 1) // pool.numel())[:n].reshape(shape)
        at = self.pool_at if self.pool_at + n <= pool.numel() else 0
        self.pool_at = at + n
        return pool[at:at + n].reshape(shape)

    def _out(self, t):
        return t.to(device=self.device, dtype=self.dtype)

    def uniform(self, shape, bound):
        if self.meta:
            return torch.empty(shape, device="meta", dtype=self.dtype)
        if self.dg is not None:
            return ((torch.rand(shape, generator=self.dg, device=self.device) * 2 - 1) * bound).to(self.dtype)
        if self.pool_u is not None:
            return self._out(self._from_pool(self.pool_u, shape) * bound)
        return self._out((torch.rand(shape, generator=self.g) * 2 - 1) * bound)

    def normal(self, shape, std, mean=0.0):
        if self.meta:
            return torch.empty(shape, device="meta", dtype=self.dtype)
        if self.dg is not None:
            return (torch.randn(shape, generator=self.dg, device=self.device) * std + mean).to(self.dtype)
        if self.pool_n is not None:
            return self._out(self._from_pool(self.pool_n, shape) * std + mean)
        return self._out(torch.randn(shape, generator=self.g) * std + mean)

    def linear(self, p, cin, cout, bias=True, small=False):
        b = 1.0 / math.sqrt(cin)
        self.sd[p + ".weight"] = self.normal((cout, cin), 0.02) if small else self.uniform((cout, cin), b)
        if bias:
            self.sd[p + ".bias"] = self.normal((cout,), 0.02) if small else self.uniform((cout,), b)

    def conv(self, p, cin, cout, k):
        b = 1.0 / math.sqrt(cin * k * k)
        self.sd[p + ".weight"] = self.uniform((cout, cin, k, k), b)
        self.sd[p + ".bias"] = self.uniform((cout,), b)

    def norm(self, p, c):
        self.sd[p + ".weight"] = self.normal((c,), 0.1, 1.0)
        self.sd[p + ".bias"] = self.normal((c,), 0.1)

    def attn(self, p, c, ctx=None, out_small=False, qkv_bias=False):
        ctx = ctx or c
        self.linear(p + ".to_q", c, c, bias=qkv_bias)
        self.linear(p + ".to_k", ctx, c, bias=qkv_bias)
        self.linear(p + ".to_v", ctx, c, bias=qkv_bias)
        self.linear(p + ".to_out.0", c, c, small=out_small)
        if out_small:   # only the weight is zero-initialised in the reference; bias keeps default init
            self.sd[p + ".to_out.0.bias"] = self.uniform((c,), 1.0 / math.sqrt(c))

    def ff(self, p, c):
        self.linear(p + ".net.0.proj", c, 8 * c)
        self.linear(p + ".net.2", 4 * c, c)

    def resnet(self, p, cin, cout, temb):
        self.norm(p + ".norm1", cin)
        self.conv(p + ".conv1", cin, cout, 3)
        if temb:
            self.linear(p + ".time_emb_proj", temb, cout)
        self.norm(p + ".norm2", cout)
        self.conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            self.conv(p + ".conv_shortcut", cin, cout, 1)


def pe_table(max_len, d_model):
    """PositionalEncoding buffer (modules/motion_module.py:262-277): pe[0,:,0::2]=sin, 1::2=cos."""
    position = torch.arange(max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def block_plan(cfg: UNetConfig):
    """Channel walk shared by both UNets (modules/unet_3d.py:112-227)."""
    ch = cfg.block_out_channels
    n = len(ch)
    plan = {"down": [], "up": []}
    out_c = ch[0]
    for i in range(n):
        in_c, out_c = out_c, ch[i]
        layers = [dict(cin=in_c if j == 0 else out_c, cout=out_c) for j in range(cfg.layers_per_block)]
        plan["down"].append(dict(prefix=f"down_blocks.{i}", layers=layers, attn=cfg.attn_levels[i],
                                 sampler=(i != n - 1), c=out_c))
    plan["mid"] = dict(prefix="mid_block", c=ch[-1])
    rev = list(reversed(ch))
    rev_attn = list(reversed(cfg.attn_levels))
    out_c = rev[0]
    for i in range(n):
        prev_out, out_c = out_c, rev[i]
        in_c = rev[min(i + 1, n - 1)]
        nl = cfg.layers_per_block + 1
        layers = []
        for j in range(nl):
            skip = in_c if j == nl - 1 else out_c
            rin = prev_out if j == 0 else out_c
            layers.append(dict(cin=rin + skip, cout=out_c, c_hidden=rin, c_skip=skip))
        plan["up"].append(dict(prefix=f"up_blocks.{i}", layers=layers, attn=rev_attn[i],
                               sampler=(i != n - 1), c=out_c))
    return plan


def _spatial_block_3d(g: _Gen, p, c, cfg):
    g.norm(p + ".norm", c)
    g.conv(p + ".proj_in", c, c, 1)
    g.conv(p + ".proj_out", c, c, 1)
    t = p + ".transformer_blocks.0"
    g.attn(t + ".attn1", c)
    g.norm(t + ".norm1", c)
    g.attn(t + ".attn1_5", c)
    g.norm(t + ".norm1_5", c)
    g.attn(t + ".attn2", c, ctx=cfg.cross_attention_dim, out_small=True)
    g.norm(t + ".norm2", c)
    g.ff(t + ".ff", c)
    g.norm(t + ".norm3", c)


def _motion_module(g: _Gen, p, c, cfg):
    t = p + ".temporal_transformer"
    g.norm(t + ".norm", c)
    g.linear(t + ".proj_in", c, c)
    b = t + ".transformer_blocks.0"
    for i in range(2):
        a = f"{b}.attention_blocks.{i}"
        g.attn(a, c)
        g.sd[a + ".pos_encoder.pe"] = g._out(pe_table(cfg.temporal_max_len, c))
        g.norm(f"{b}.norms.{i}", c)
    g.ff(b + ".ff", c)
    g.norm(b + ".ff_norm", c)
    g.linear(t + ".proj_out", c, c, small=True)


def _spatial_block_2d(g: _Gen, p, c, cfg):
    g.norm(p + ".norm", c)
    g.conv(p + ".proj_in", c, c, 1)
    g.conv(p + ".proj_out", c, c, 1)
    t = p + ".transformer_blocks.0"
    g.norm(t + ".norm1", c)
    g.attn(t + ".attn1", c)
    g.norm(t + ".norm2", c)
    g.attn(t + ".attn2", c, ctx=cfg.cross_attention_dim)
    g.norm(t + ".norm3", c)
    g.ff(t + ".ff", c)


def _unet_common(g: _Gen, cfg: UNetConfig, three_d: bool):
    ch0 = cfg.block_out_channels[0]
    temb = cfg.time_embed_dim
    g.conv("conv_in", cfg.in_channels, ch0, 3)
    g.linear("time_embedding.linear_1", ch0, temb)
    g.linear("time_embedding.linear_2", temb, temb)
    plan = block_plan(cfg)
    spatial = _spatial_block_3d if three_d else _spatial_block_2d
    for blk in plan["down"] + plan["up"]:
        p = blk["prefix"]
        for j, l in enumerate(blk["layers"]):
            g.resnet(f"{p}.resnets.{j}", l["cin"], l["cout"], temb)
            if blk["attn"]:
                spatial(g, f"{p}.attentions.{j}", l["cout"], cfg)
            if three_d:
                _motion_module(g, f"{p}.motion_modules.{j}", l["cout"], cfg)
        if blk["sampler"]:
            name = "downsamplers" if p.startswith("down") else "upsamplers"
            g.conv(f"{p}.{name}.0.conv", blk["c"], blk["c"], 3)
    c = plan["mid"]["c"]
    g.resnet("mid_block.resnets.0", c, c, temb)
    spatial(g, "mid_block.attentions.0", c, cfg)
    if three_d:
        _motion_module(g, "mid_block.motion_modules.0", c, cfg)
    g.resnet("mid_block.resnets.1", c, c, temb)
    if three_d:
        g.norm("conv_norm_out", ch0)
    g.conv("conv_out", ch0, cfg.out_channels, 3)


def _gen(seed, device, dtype, draw_on_device, timing_only=False):
    g = _Gen(seed, device, dtype)
    if timing_only:
        g.use_pool()
    if draw_on_device and str(device) not in ("cpu", "meta"):
        g.dg = torch.Generator(device=device)
        g.dg.manual_seed(seed)
    return g


def unet3d_state_dict(cfg: UNetConfig = None, seed=42, device="cpu", dtype=torch.float32, draw_on_device=False, timing_only=False):
    """Denoising UNet3DConditionModel weights (1386 tensors at the SD-1.5 config)."""
    g = _gen(seed, device, dtype, draw_on_device, timing_only)
    _unet_common(g, cfg or UNetConfig(), True)
    return g.sd


def refnet_state_dict(cfg: UNetConfig = None, seed=43, device="cpu", dtype=torch.float32, draw_on_device=False):
    """ReferenceNet (UNet2DConditionModel, conv_norm_out=None: unet_2d_condition.py:650) weights."""
    g = _gen(seed, device, dtype, draw_on_device)
    _unet_common(g, cfg or UNetConfig(), False)
    return g.sd


def vae_decoder_state_dict(cfg: VaeConfig = None, seed=44, device="cpu", dtype=torch.float32, draw_on_device=False, timing_only=False):
    """sd-vae-ft-mse decoder half of diffusers AutoencoderKL (post_quant_conv + decoder.*)."""
    cfg = cfg or VaeConfig()
    g = _gen(seed, device, dtype, draw_on_device, timing_only)
    ch = list(cfg.block_out_channels)
    g.conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    g.conv("decoder.conv_in", cfg.latent_channels, ch[-1], 3)
    for j in range(2):
        g.resnet(f"decoder.mid_block.resnets.{j}", ch[-1], ch[-1], None)
    a = "decoder.mid_block.attentions.0"
    g.norm(a + ".group_norm", ch[-1])
    g.attn(a, ch[-1], qkv_bias=True)
    rev = list(reversed(ch))
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            g.resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else c, c, None)
        if i != len(rev) - 1:
            g.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
        prev = c
    g.norm("decoder.conv_norm_out", ch[0])
    g.conv("decoder.conv_out", ch[0], cfg.out_channels, 3)
    return g.sd


def synthetic_inputs(cfg: UNetConfig, num_frames, latent_h, latent_w, seed=42, device="cpu", dtype=torch.float32):
    """Synthetic clip inputs: start latents N(0,1) drawn in fp32 on CPU then cast;
    reference latent N(0,1)*0.18215; kps features cat([0, N(0,0.1^2)]); audio cat([0, N(0,1)])."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    c0 = cfg.block_out_channels[0]
    latents = torch.randn(1, cfg.in_channels, num_frames, latent_h, latent_w, generator=g)
    ref_latents = torch.randn(1, cfg.in_channels, latent_h, latent_w, generator=g) * 0.18215
    kps = torch.randn(1, c0, num_frames, latent_h, latent_w, generator=g) * 0.1
    kps = torch.cat([torch.zeros_like(kps), kps], dim=0)
    audio = torch.randn(1, num_frames, 5, cfg.cross_attention_dim, generator=g)
    audio = torch.cat([torch.zeros_like(audio), audio], dim=0)
    to = dict(device=device, dtype=dtype)
    return dict(latents=latents.to(**to), ref_latents=ref_latents.to(**to), kps_features=kps.to(**to),
                audio_embeddings=audio.to(**to))


# ---------------------------------------------------------------------------------------------------------------
# Once-per-clip prologue models: VKpsGuider, AudioProjection, VAE encoder.
@dataclass
class KpsGuiderConfig:
    """modules/v_kps_guider.py:11-16 as instantiated by inference.py:100."""
    conditioning_embedding_channels: int = 320
    conditioning_channels: int = 3
    block_out_channels: Tuple[int, ...] = (16, 32, 96, 256)


@dataclass
class AudioProjectionConfig:
    """modules/audio_projection.py:98-110 as instantiated by inference.py:116-126 (num_pad_audio_frames = 2)."""
    dim: int = 768
    depth: int = 4
    dim_head: int = 64
    heads: int = 12
    num_queries: int = 5
    embedding_dim: int = 768
    output_dim: int = 768
    ff_mult: int = 4
    max_seq_len: int = 10


@dataclass
class Wav2Vec2Config:
    """facebook/wav2vec2-base-960h (the audio encoder inference.py:165-166 loads): transformers Wav2Vec2Config fields
    that shape the eval-mode forward.  Only the group-norm / post-LayerNorm variant is supported."""
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    conv_dim: Tuple[int, ...] = (512, 512, 512, 512, 512, 512, 512)
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    feat_extract_norm: str = "group"
    do_stable_layer_norm: bool = False
    conv_bias: bool = False

    def num_frames(self, samples):
        """time steps the feature encoder produces for `samples` waveform samples"""
        for k, s in zip(self.conv_kernel, self.conv_stride):
            samples = (samples - k) // s + 1
        return samples


def wav2vec2_state_dict(cfg: Wav2Vec2Config = None, seed=48, device="cpu", dtype=torch.float32):
    """transformers Wav2Vec2Model state_dict (key names of transformers >= 4.3x with torch parametrized weight_norm).
    Convolutions are drawn Kaiming-normal like the library's own init so that seven GELU layers keep a live signal."""
    cfg = cfg or Wav2Vec2Config()
    g = _gen(seed, device, dtype, False)
    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        p = f"feature_extractor.conv_layers.{i}"
        g.sd[p + ".conv.weight"] = g.normal((c, cin, k), math.sqrt(2.0 / (cin * k)))
        if i == 0:
            g.norm(p + ".layer_norm", c)
        cin = c
    h = cfg.hidden_size
    g.sd["masked_spec_embed"] = g.uniform((h,), 1.0)
    g.norm("feature_projection.layer_norm", cin)
    g.linear("feature_projection.projection", cin, h)
    kp, gp = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    p = "encoder.pos_conv_embed.conv"
    g.sd[p + ".bias"] = g.normal((h,), 0.02)
    v = g.normal((h, h // gp, kp), 2.0 * math.sqrt(1.0 / (kp * h)))
    g.sd[p + ".parametrizations.weight.original0"] = (v.float().norm(dim=(0, 1), keepdim=True) *
                                                      (1.0 + 0.1 * g.normal((1, 1, kp), 1.0).float())).to(v.dtype)
    g.sd[p + ".parametrizations.weight.original1"] = v
    g.norm("encoder.layer_norm", h)
    for i in range(cfg.num_hidden_layers):
        lp = f"encoder.layers.{i}"
        for n in ("k", "v", "q", "out"):
            g.linear(f"{lp}.attention.{n}_proj", h, h)
        g.norm(lp + ".layer_norm", h)
        g.linear(lp + ".feed_forward.intermediate_dense", h, cfg.intermediate_size)
        g.linear(lp + ".feed_forward.output_dense", cfg.intermediate_size, h)
        g.norm(lp + ".final_layer_norm", h)
    return g.sd


def kps_guider_state_dict(cfg: KpsGuiderConfig = None, seed=45, device="cpu", dtype=torch.float32):
    """VKpsGuider state_dict (modules/v_kps_guider.py:18-33).  conv_out is zero-initialised in the reference
    (zero_module); here it is drawn N(0, 0.02^2) so the path is numerically live."""
    cfg = cfg or KpsGuiderConfig()
    g = _gen(seed, device, dtype, False)
    ch = list(cfg.block_out_channels)
    g.conv("conv_in", cfg.conditioning_channels, ch[0], 3)
    for i in range(len(ch) - 1):
        g.conv(f"blocks.{2 * i}", ch[i], ch[i], 3)
        g.conv(f"blocks.{2 * i + 1}", ch[i], ch[i + 1], 3)
    g.sd["conv_out.weight"] = g.normal((cfg.conditioning_embedding_channels, ch[-1], 3, 3), 0.02)
    g.sd["conv_out.bias"] = g.normal((cfg.conditioning_embedding_channels,), 0.02)
    return g.sd


def audio_projection_state_dict(cfg: AudioProjectionConfig = None, seed=46, device="cpu", dtype=torch.float32):
    """AudioProjection state_dict (modules/audio_projection.py:112-136)."""
    cfg = cfg or AudioProjectionConfig()
    g = _gen(seed, device, dtype, False)
    inner = cfg.dim_head * cfg.heads
    g.sd["pos_emb.weight"] = g.normal((cfg.max_seq_len, cfg.embedding_dim), 1.0)
    g.sd["latents"] = g.normal((1, cfg.num_queries, cfg.dim), cfg.dim ** -0.5)
    g.linear("proj_in", cfg.embedding_dim, cfg.dim)
    g.linear("proj_out", cfg.dim, cfg.output_dim)
    g.norm("norm_out", cfg.output_dim)
    for i in range(cfg.depth):
        a, f = f"layers.{i}.0", f"layers.{i}.1"
        g.norm(a + ".norm1", cfg.dim)
        g.norm(a + ".norm2", cfg.dim)
        g.linear(a + ".to_q", cfg.dim, inner, bias=False)
        g.linear(a + ".to_kv", cfg.dim, 2 * inner, bias=False)
        g.linear(a + ".to_out", inner, cfg.dim, bias=False)
        g.norm(f + ".0", cfg.dim)
        g.linear(f + ".1", cfg.dim, cfg.dim * cfg.ff_mult, bias=False)
        g.linear(f + ".3", cfg.dim * cfg.ff_mult, cfg.dim, bias=False)
    return g.sd


def vae_encoder_state_dict(cfg: VaeConfig = None, seed=47, device="cpu", dtype=torch.float32):
    """sd-vae-ft-mse encoder half of diffusers AutoencoderKL (encoder.* + quant_conv)."""
    cfg = cfg or VaeConfig()
    g = _gen(seed, device, dtype, False)
    ch = list(cfg.block_out_channels)
    g.conv("encoder.conv_in", 3, ch[0], 3)
    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(cfg.layers_per_block):
            g.resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c, None)
        if i != len(ch) - 1:
            g.conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
        prev = c
    for j in range(2):
        g.resnet(f"encoder.mid_block.resnets.{j}", ch[-1], ch[-1], None)
    a = "encoder.mid_block.attentions.0"
    g.norm(a + ".group_norm", ch[-1])
    g.attn(a, ch[-1], qkv_bias=True)
    g.norm("encoder.conv_norm_out", ch[-1])
    g.conv("encoder.conv_out", ch[-1], 2 * cfg.latent_channels, 3)
    g.conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    return g.sd
