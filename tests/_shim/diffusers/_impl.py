"""Leaf classes of the diffusers 0.29.2 stand-in (see __init__.py).  TEST INFRASTRUCTURE ONLY.

Written from the published behaviour of diffusers 0.29.2; no
diffusers source is available offline.  Only what /root/reference executes is real; the
rest are inert stubs that raise if ever instantiated.
"""
import functools
import inspect
import json
import math
from collections import OrderedDict
from dataclasses import fields, is_dataclass

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- utils
class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


class logging:  # noqa: N801  (mirrors diffusers.utils.logging module API)
    @staticmethod
    def get_logger(name=None):
        return _Logger()


def deprecate(*args, **kwargs):
    return None


def is_torch_version(op, ver):
    return True


def make_stub(name):
    def __init__(self, *a, **k):
        raise NotImplementedError(f"diffusers shim: {name} is a never-executed import stub")

    return type(name, (nn.Module,), {"__init__": __init__})


class BaseOutput(OrderedDict):
    """Dataclass-friendly ordered dict with attribute access (diffusers.utils.BaseOutput)."""

    def __post_init__(self):
        if is_dataclass(self):
            for f in fields(self):
                v = getattr(self, f.name)
                if v is not None:
                    OrderedDict.__setitem__(self, f.name, v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return OrderedDict.__getitem__(self, k)
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    device = device or torch.device("cpu")
    gen_device = generator.device if generator is not None else device
    if isinstance(generator, list):
        raise NotImplementedError
    t = torch.randn(shape, generator=generator, device=gen_device, dtype=dtype)
    return t.to(device)


# ----------------------------------------------------------------------------- config / model mixins
class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def register_to_config(init):
    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = [p for n, p in sig.parameters.items() if n != "self"]
        cfg = {p.name: p.default for p in params if p.default is not inspect.Parameter.empty}
        for p, a in zip(params, args):
            cfg[p.name] = a
        init_kwargs = {k: v for k, v in kwargs.items() if not k.startswith("_")}
        cfg.update(init_kwargs)
        init(self, *args, **init_kwargs)
        if not hasattr(self, "_internal_dict"):
            self._internal_dict = FrozenDict()
        self._internal_dict = FrozenDict({**self._internal_dict, **cfg})

    return wrapper


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kwargs):
        prev = getattr(self, "_internal_dict", {})
        self._internal_dict = FrozenDict({**prev, **kwargs})

    @classmethod
    def load_config(cls, path, **kwargs):
        if isinstance(path, dict):
            return dict(path)
        with open(path) as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config, **kwargs):
        if not isinstance(config, dict):
            config = cls.load_config(config)
        sig = inspect.signature(cls.__init__)
        accepted = set(sig.parameters) - {"self"}
        init = {k: v for k, v in config.items() if k in accepted}
        init.update({k: v for k, v in kwargs.items() if k in accepted})
        return cls(**init)



class ModelMixin(nn.Module):
    _supports_gradient_checkpointing = False

    def __getattr__(self, name):
        # diffusers lets `model.foo` fall back to `model.config.foo` (with a deprecation warning)
        d = self.__dict__.get("_internal_dict")
        if d is not None and name in d:
            return d[name]
        return super().__getattr__(name)

    @property
    def dtype(self):
        for p in self.parameters():
            return p.dtype
        return torch.float32

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")


class UNet2DConditionLoadersMixin:
    pass


# ----------------------------------------------------------------------------- activations / embeddings
def get_activation(name):
    name = name.lower()
    if name in ("swish", "silu"):
        return nn.SiLU()
    if name == "mish":
        return nn.Mish()
    if name == "gelu":
        return nn.GELU()
    if name == "relu":
        return nn.ReLU()
    raise ValueError(name)


def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1.0,
                           scale=1.0, max_period=10000):
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half_dim, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos,
                                      self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None,
                 cond_proj_dim=None, sample_proj_bias=True):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim, sample_proj_bias)
        self.act = get_activation(act_fn)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim, sample_proj_bias)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


# ----------------------------------------------------------------------------- attention / feed-forward
class AttnProcessor2_0:
    """softmax(q k^T / sqrt(d)) v via F.scaled_dot_product_attention, then to_out."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 *args, **kwargs):
        residual = hidden_states
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, h, w = hidden_states.shape
            hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
        batch = hidden_states.shape[0]
        if attn.group_norm is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        q = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        k = attn.to_k(encoder_hidden_states)
        v = attn.to_v(encoder_hidden_states)
        inner = k.shape[-1]
        hd = inner // attn.heads
        q = q.view(batch, -1, attn.heads, hd).transpose(1, 2)
        k = k.view(batch, -1, attn.heads, hd).transpose(1, 2)
        v = v.view(batch, -1, attn.heads, hd).transpose(1, 2)
        if attention_mask is not None:
            raise NotImplementedError("shim: attention_mask is never used on the reference hot path")
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(batch, -1, attn.heads * hd).to(q.dtype)
        o = attn.to_out[0](o)
        o = attn.to_out[1](o)
        if input_ndim == 4:
            o = o.transpose(-1, -2).reshape(b, c, h, w)
        if attn.residual_connection:
            o = o + residual
        return o / attn.rescale_output_factor


class AttnProcessor(AttnProcessor2_0):
    """Classic (baddbmm + softmax) processor: same arithmetic as the SDPA one up to round-off."""


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None,
                 cross_attention_norm_num_groups=32, qk_norm=None, added_kv_proj_dim=None, norm_num_groups=None,
                 spatial_norm_dim=None, out_bias=True, scale_qk=True, only_cross_attention=False, eps=1e-5,
                 rescale_output_factor=1.0, residual_connection=False, _from_deprecated_attn_block=False,
                 processor=None, out_dim=None, context_pre_only=None):
        super().__init__()
        self.inner_dim = out_dim if out_dim is not None else dim_head * heads
        self.query_dim = query_dim
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.heads = heads
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.only_cross_attention = only_cross_attention
        self.group_norm = (nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True)
                           if norm_num_groups is not None else None)
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.processor = processor if processor is not None else AttnProcessor2_0()

    def set_processor(self, processor):
        self.processor = processor

    def set_use_memory_efficient_attention_xformers(self, *a, **k):
        pass

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)

    def forward(self, hidden_states, *args, **kwargs):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False,
                 inner_dim=None, bias=True):
        super().__init__()
        inner_dim = inner_dim or int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        assert activation_fn == "geglu"
        self.net = nn.ModuleList([GEGLU(dim, inner_dim, bias=bias), nn.Dropout(dropout),
                                  nn.Linear(inner_dim, dim_out, bias=bias)])

    def forward(self, hidden_states, *args, **kwargs):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


# ----------------------------------------------------------------------------- 2-D resnet / resampling
class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv",
                 kernel_size=None, padding=1, **kw):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.name = name
        assert use_conv and not use_conv_transpose
        conv = nn.Conv2d(self.channels, self.out_channels, kernel_size or 3, padding=padding)
        if name == "conv":
            self.conv = conv
        else:
            self.Conv2d_0 = conv

    def forward(self, hidden_states, output_size=None, *args, **kwargs):
        dtype = hidden_states.dtype
        if dtype == torch.bfloat16:
            hidden_states = hidden_states.to(torch.float32)
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        if dtype == torch.bfloat16:
            hidden_states = hidden_states.to(dtype)
        conv = self.conv if self.name == "conv" else self.Conv2d_0
        return conv(hidden_states)


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv", kernel_size=3, **kw):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.padding = padding
        assert use_conv
        conv = nn.Conv2d(self.channels, self.out_channels, kernel_size, stride=2, padding=padding)
        if name == "conv":
            self.Conv2d_0 = conv
            self.conv = conv
        elif name == "Conv2d_0":
            self.conv = conv
        else:
            self.conv = conv

    def forward(self, hidden_states, *args, **kwargs):
        if self.use_conv and self.padding == 0:
            hidden_states = F.pad(hidden_states, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(hidden_states)


class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512,
                 groups=32, groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish", skip_time_act=False,
                 time_embedding_norm="default", kernel=None, output_scale_factor=1.0, use_in_shortcut=None,
                 up=False, down=False, conv_shortcut_bias=True, conv_2d_out_channels=None):
        super().__init__()
        assert time_embedding_norm == "default" and not up and not down
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.output_scale_factor = output_scale_factor
        groups_out = groups_out or groups
        self.norm1 = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(num_groups=groups_out, num_channels=out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        conv_2d_out_channels = conv_2d_out_channels or out_channels
        self.conv2 = nn.Conv2d(out_channels, conv_2d_out_channels, 3, stride=1, padding=1)
        self.nonlinearity = get_activation(non_linearity)
        self.use_in_shortcut = (self.in_channels != conv_2d_out_channels if use_in_shortcut is None
                                else use_in_shortcut)
        self.conv_shortcut = None
        if self.use_in_shortcut:
            self.conv_shortcut = nn.Conv2d(in_channels, conv_2d_out_channels, 1, stride=1, padding=0,
                                           bias=conv_shortcut_bias)

    def forward(self, input_tensor, temb=None, *args, **kwargs):
        h = self.norm1(input_tensor)
        h = self.nonlinearity(h)
        h = self.conv1(h)
        if self.time_emb_proj is not None:
            t = self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
            h = h + t
        h = self.norm2(h)
        h = self.nonlinearity(h)
        h = self.dropout(h)
        h = self.conv2(h)
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


# ----------------------------------------------------------------------------- DDIM scheduler
class _SchedOut(BaseOutput):
    pass


def _rescale_zero_terminal_snr(betas):
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    alphas_bar_sqrt = alphas_cumprod.sqrt()
    a0 = alphas_bar_sqrt[0].clone()
    aT = alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt -= aT
    alphas_bar_sqrt *= a0 / (a0 - aT)
    alphas_bar = alphas_bar_sqrt ** 2
    alphas = alphas_bar[1:] / alphas_bar[:-1]
    alphas = torch.cat([alphas_bar[0:1], alphas])
    return 1 - alphas


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False):
        self.config = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                 beta_schedule=beta_schedule, clip_sample=clip_sample,
                                 set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                                 prediction_type=prediction_type, thresholding=thresholding,
                                 clip_sample_range=clip_sample_range, timestep_spacing=timestep_spacing,
                                 rescale_betas_zero_snr=rescale_betas_zero_snr)
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        if rescale_betas_zero_snr:
            self.betas = _rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        T = self.config.num_train_timesteps
        sp = self.config.timestep_spacing
        if sp == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif sp == "leading":
            ratio = T // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        elif sp == "trailing":
            ratio = T / num_inference_steps
            ts = np.round(np.arange(T, 0, -ratio)).astype(np.int64)
            ts -= 1
        else:
            raise ValueError(sp)
        self.timesteps = torch.from_numpy(ts).to(device)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        prev_t = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif pt == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        elif pt == "v_prediction":
            x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
            eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        else:
            raise ValueError(pt)
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
        std = eta * var ** 0.5
        assert eta == 0.0, "shim: only the deterministic (eta=0) DDIM update is restated"
        direction = (1 - a_prev - std ** 2) ** 0.5 * eps
        prev_sample = a_prev ** 0.5 * x0 + direction
        out = _SchedOut()
        out["prev_sample"] = prev_sample
        out["pred_original_sample"] = x0
        out.prev_sample = prev_sample
        out.pred_original_sample = x0
        return out


# ----------------------------------------------------------------------------- AutoencoderKL (sd-vae-ft-mse layout)
class _VaeMid(nn.Module):
    def __init__(self, ch, groups, eps=1e-6):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=ch, out_channels=ch, temb_channels=None, groups=groups, eps=eps),
            ResnetBlock2D(in_channels=ch, out_channels=ch, temb_channels=None, groups=groups, eps=eps)])
        self.attentions = nn.ModuleList([
            Attention(ch, heads=1, dim_head=ch, rescale_output_factor=1.0, eps=eps, norm_num_groups=groups,
                      residual_connection=True, bias=True, upcast_softmax=True, _from_deprecated_attn_block=True)])

    def forward(self, x):
        x = self.resnets[0](x, None)
        x = self.attentions[0](x)
        return self.resnets[1](x, None)


class _UpDecoderBlock(nn.Module):
    def __init__(self, cin, cout, n, add_upsample, groups, eps=1e-6):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=cin if i == 0 else cout, out_channels=cout, temb_channels=None,
                          groups=groups, eps=eps) for i in range(n)])
        self.upsamplers = (nn.ModuleList([Upsample2D(cout, use_conv=True, out_channels=cout)])
                           if add_upsample else None)

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class _DownEncoderBlock(nn.Module):
    def __init__(self, cin, cout, n, add_downsample, groups, eps=1e-6):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=cin if i == 0 else cout, out_channels=cout, temb_channels=None,
                          groups=groups, eps=eps) for i in range(n)])
        self.downsamplers = (nn.ModuleList([Downsample2D(cout, use_conv=True, out_channels=cout, padding=0,
                                                         name="op")]) if add_downsample else None)

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class _Decoder(nn.Module):
    def __init__(self, latent, out_ch, chans, layers, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(latent, chans[-1], 3, padding=1)
        self.mid_block = _VaeMid(chans[-1], groups)
        rev = list(reversed(chans))
        ups, prev = [], rev[0]
        for i, c in enumerate(rev):
            ups.append(_UpDecoderBlock(prev, c, layers + 1, i != len(rev) - 1, groups))
            prev = c
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(num_channels=chans[0], num_groups=groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(chans[0], out_ch, 3, padding=1)

    def forward(self, z):
        x = self.conv_in(z)
        x = self.mid_block(x)
        for u in self.up_blocks:
            x = u(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class _Encoder(nn.Module):
    def __init__(self, in_ch, latent, chans, layers, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(in_ch, chans[0], 3, padding=1)
        downs, prev = [], chans[0]
        for i, c in enumerate(chans):
            downs.append(_DownEncoderBlock(prev, c, layers, i != len(chans) - 1, groups))
            prev = c
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = _VaeMid(chans[-1], groups)
        self.conv_norm_out = nn.GroupNorm(num_channels=chans[-1], num_groups=groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(chans[-1], 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for d in self.down_blocks:
            x = d(x)
        x = self.mid_block(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class _Dist:
    def __init__(self, moments):
        self.mean, self.logvar = moments.chunk(2, dim=1)


class _Holder:
    pass


class AutoencoderKL(ModelMixin, ConfigMixin):
    @register_to_config
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, sample_size=256, scaling_factor=0.18215):
        super().__init__()
        self.encoder = _Encoder(in_channels, latent_channels, list(block_out_channels), layers_per_block,
                                norm_num_groups)
        self.decoder = _Decoder(latent_channels, out_channels, list(block_out_channels), layers_per_block,
                                norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    def encode(self, x):
        h = _Holder()
        h.latent_dist = _Dist(self.quant_conv(self.encoder(x)))
        return h

    def decode(self, z):
        h = _Holder()
        h.sample = self.decoder(self.post_quant_conv(z))
        return h


# ----------------------------------------------------------------------------- pipeline plumbing
class _Bar:
    def __init__(self, total=None):
        self.total = total

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, n=1):
        pass

    def set_description(self, s):
        pass


class DiffusionPipeline:
    def register_modules(self, **kwargs):
        self._modules_ = dict(kwargs)
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to(self, *args, **kwargs):
        for v in self._modules_.values():
            if isinstance(v, nn.Module):
                v.to(*args, **kwargs)
        return self

    @property
    def device(self):
        for v in self._modules_.values():
            if isinstance(v, nn.Module):
                for p in v.parameters():
                    return p.device
        return torch.device("cpu")

    @property
    def dtype(self):
        for v in self._modules_.values():
            if isinstance(v, nn.Module):
                for p in v.parameters():
                    return p.dtype
        return torch.float32

    def progress_bar(self, iterable=None, total=None):
        return _Bar(total)


class VaeImageProcessor:
    def __init__(self, do_resize=True, vae_scale_factor=8, resample="lanczos", do_normalize=True,
                 do_binarize=False, do_convert_rgb=False, do_convert_grayscale=False):
        self.vae_scale_factor = vae_scale_factor
        self.do_normalize = do_normalize
        self.do_convert_rgb = do_convert_rgb

    def preprocess(self, image, height=None, width=None):
        from PIL import Image
        if isinstance(image, torch.Tensor):
            t = image if image.ndim == 4 else image[None]
        else:
            if self.do_convert_rgb:
                image = image.convert("RGB")
            if height is not None and width is not None:
                image = image.resize((width, height), resample=Image.LANCZOS)
            arr = np.array(image).astype(np.float32) / 255.0
            t = torch.from_numpy(arr).permute(2, 0, 1)[None]
        if self.do_normalize:
            t = 2.0 * t - 1.0
        return t
