"""wav2vec2 audio encoder on libvexpress_hip kernels ("wav2vec2 -> interpolation/windowing").

Drop-in for the `audio_encoder` the reference pipeline calls once per clip (`self.audio_encoder(wave).last_hidden_state`,
pipelines/v_express_pipeline.py:377; transformers `Wav2Vec2Model.from_pretrained("facebook/wav2vec2-base-960h")`,
inference.py:165-166), for the one variant that checkpoint uses: group-norm feature encoder, post-LayerNorm
transformer, eval mode.  Same constructor-from-config / `load_state_dict` / `forward(input_values)` surface and the
transformers state_dict key names.

Everything runs time-major, `[T, C]` bf16 tokens, on the hot path's own kernels:
  * conv layer 0 (1 input channel, raw float32 waveform)  -> `vx_wave_conv1d`, then `vx_groupnorm` with one group per
    channel over the time axis and the erf-GELU fused into its apply pass;
  * conv layers 1..6: in the time-major layout a k-tap, stride-s window is k*C CONTIGUOUS elements, so each layer is
    one `vx_gemm` over overlapping rows (row stride s*C < row length k*C) - no im2col copy - with GELU in the epilogue;
  * positional conv (k = 128, 16 groups): the tokens are re-laid out group-major with the zero padding in place, then
    one overlapping-row `vx_gemm` per group (K = 128 * 48) with bias + GELU + the residual add fused;
  * 12 post-LN layers: fused QKV GEMM emitting V^T, flash attention (d = 64), out-projection with fused residual,
    `vx_layernorm`, GELU-fused feed-forward.
"""
import json
import os
from types import SimpleNamespace

import torch

from . import blocks as B
from . import lib as L
from . import ops
from . import weights as Wt
from .prologue import _Module
from .synth import Wav2Vec2Config


class WaveformProcessor:
    """The part of transformers' Wav2Vec2Processor the pipeline uses (pipelines/v_express_pipeline.py:375):
    `processor(wave, return_tensors="pt", sampling_rate=16000)["input_values"]` -> zero-mean / unit-variance
    float32 `[1, samples]` (Wav2Vec2FeatureExtractor with do_normalize=True, eps 1e-7)."""

    def __init__(self, sampling_rate=16000, do_normalize=True):
        self.sampling_rate, self.do_normalize = sampling_rate, do_normalize

    def __call__(self, raw_speech, return_tensors="pt", sampling_rate=None, **kwargs):
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(f"the audio encoder expects {self.sampling_rate} Hz input, got {sampling_rate}")
        wav = torch.as_tensor(raw_speech, dtype=torch.float32).reshape(1, -1)
        if self.do_normalize:
            wav = (wav - wav.mean()) / torch.sqrt(wav.var(unbiased=False) + 1e-7)
        return {"input_values": wav}


class Wav2Vec2Model(_Module):
    wants_fp32_input = True          # the pipeline hands the normalised waveform over in float32, not the compute dtype

    def __init__(self, config=None, **kwargs):
        super().__init__()
        if config is None:
            config = Wav2Vec2Config(**kwargs)
        elif not isinstance(config, Wav2Vec2Config):          # a transformers Wav2Vec2Config or a plain dict
            src = config if isinstance(config, dict) else config.to_dict()
            config = Wav2Vec2Config(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in src.items()
                                       if k in Wav2Vec2Config.__dataclass_fields__})
        if config.feat_extract_norm != "group" or config.do_stable_layer_norm:
            raise NotImplementedError("only the wav2vec2-base variant V-Express loads is implemented "
                                      "(feat_extract_norm='group', do_stable_layer_norm=False)")
        if len(set(config.conv_dim)) != 1 or config.conv_dim[0] % 8 or config.conv_kernel[0] > 16:
            raise NotImplementedError("feature-encoder widths must be equal multiples of 8, first kernel <= 16 taps")
        self.config = self.cfg = config

    @classmethod
    def from_pretrained(cls, path, dtype=torch.bfloat16, device="cuda"):
        """`Wav2Vec2Model.from_pretrained(audio_encoder_path)` (inference.py:165) for the transformers directory layout:
        `config.json` + `model.safetensors` / `pytorch_model.bin`.  A `Wav2Vec2ForCTC` checkpoint (what
        wav2vec2-base-960h is) is accepted: the `wav2vec2.` prefix is stripped and the CTC head dropped."""
        from .checkpoints import _load_file
        with open(os.path.join(path, "config.json")) as f:
            model = cls(json.load(f))
        for name in ("model.safetensors", "pytorch_model.bin"):
            if os.path.exists(os.path.join(path, name)):
                sd = _load_file(os.path.join(path, name))
                break
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
        model.load_state_dict(sd)
        return model.to(dtype=dtype, device=device)

    def _schema(self):
        from . import synth
        sd = synth.wav2vec2_state_dict(self.cfg, device="meta")
        sd.pop("masked_spec_embed", None)                      # training-time SpecAugment vector: never read
        return sd

    def load_state_dict(self, sd, strict=True):
        """Accepts the three spellings of the weight-normalised positional conv that transformers / torch versions
        have written (`weight_g`/`weight_v`, `parametrizations.weight.original0/1`, or a plain materialised `weight`)
        and stores the parametrized form."""
        out = {}
        pc = "encoder.pos_conv_embed.conv."
        for k, v in sd.items():
            k = k[len("wav2vec2."):] if k.startswith("wav2vec2.") else k
            if k.startswith(("lm_head.", "quantizer.", "project_q.", "project_hid.")) or k == "masked_spec_embed":
                continue
            if k == pc + "weight_g":
                k = pc + "parametrizations.weight.original0"
            elif k == pc + "weight_v":
                k = pc + "parametrizations.weight.original1"
            elif k == pc + "weight":
                out[pc + "parametrizations.weight.original0"] = v.float().norm(dim=(0, 1), keepdim=True).to(v.dtype)
                k = pc + "parametrizations.weight.original1"
            out[k] = v
        return super().load_state_dict(out, strict)

    # ------------------------------------------------------------------ weights
    def _pos_conv_weight(self):
        sd, p = self._raw, "encoder.pos_conv_embed.conv."
        g, v = sd[p + "parametrizations.weight.original0"], sd[p + "parametrizations.weight.original1"]
        g, v = g.float(), v.float()
        return g * v / v.norm(dim=(0, 1), keepdim=True)            # weight_norm(dim=2): one norm per tap

    def _prepared(self):
        if self._P is not None:
            return self._P
        self._need_gpu()
        sd, dev, cfg = self._raw, self._device, self.cfg
        f32 = dict(device=dev, dtype=torch.float32)
        fe = "feature_extractor.conv_layers."
        # layer 0: [C, 1, taps] -> [taps, C] float32.  A conv bias here would be removed exactly by the per-channel
        # GroupNorm that follows (it normalises every channel over time), so it is not needed.
        P = Wt.Prepared(conv0=sd[fe + "0.conv.weight"][:, 0, :].t().to(**f32).contiguous(),
                        gn0=Wt.prep_norm(sd, fe + "0.layer_norm", dev), convs=[], layers=[])
        for i in range(1, len(cfg.conv_dim)):
            w = sd[fe + f"{i}.conv.weight"]                                          # [Cout, Cin, k]
            b = sd.get(fe + f"{i}.conv.bias")
            P.convs.append((Wt._dev(w.permute(0, 2, 1).reshape(w.shape[0], -1), dev, Wt.BF16),
                            None if b is None else b.to(**f32).contiguous(), w.shape[2], cfg.conv_stride[i]))
        P["fp_norm"] = Wt.prep_norm(sd, "feature_projection.layer_norm", dev)
        P["fp"] = Wt.prep_linear(sd, "feature_projection.projection", dev)
        wp = self._pos_conv_weight()                                                 # [H, H/G, k]
        G = cfg.num_conv_pos_embedding_groups
        cg = wp.shape[0] // G
        if cg % 8:
            raise NotImplementedError(f"positional-conv group width {cg} must be a multiple of 8")
        P["pos_w"] = [Wt._dev(wp[g * cg:(g + 1) * cg].permute(0, 2, 1).reshape(cg, -1), dev, Wt.BF16) for g in range(G)]
        P["pos_b"] = sd["encoder.pos_conv_embed.conv.bias"].to(**f32).contiguous()
        P["enc_norm"] = Wt.prep_norm(sd, "encoder.layer_norm", dev)
        for i in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{i}."
            qkv = [p + f"attention.{n}_proj" for n in ("q", "k", "v")]
            attn = Wt.Prepared(wqkv=Wt._cat_w(sd, [k + ".weight" for k in qkv], dev),
                               bqkv=Wt._cat_b(sd, [k + ".bias" for k in qkv], dev),
                               out=Wt.prep_linear(sd, p + "attention.out_proj", dev))
            P.layers.append(Wt.Prepared(attn=attn, ln1=Wt.prep_norm(sd, p + "layer_norm", dev),
                                        ff1=Wt.prep_linear(sd, p + "feed_forward.intermediate_dense", dev),
                                        ff2=Wt.prep_linear(sd, p + "feed_forward.output_dense", dev),
                                        ln2=Wt.prep_norm(sd, p + "final_layer_norm", dev)))
        self._P = P
        return P

    # ------------------------------------------------------------------ forward
    def extract_features(self, wave):
        """Wav2Vec2FeatureEncoder on one waveform: float32 [samples] -> bf16 [T, conv_dim]."""
        P, cfg = self._prepared(), self.cfg
        h = ops.wave_conv1d(wave, P.conv0, cfg.conv_stride[0])
        T, c = h.shape
        h = ops.groupnorm(h.view(1, T, c), P.gn0.g, P.gn0.b, frames=1, hw=T, groups=c, eps=1e-5,
                          silu=L.VX_ACT_GELU).view(T, c)
        for w, b, k, s in P.convs:
            t_out = (T - k) // s + 1
            if t_out < 1:
                raise ValueError("waveform too short for the feature encoder")
            win = torch.as_strided(h, (t_out, k * c), (s * c, 1), h.storage_offset())   # overlapping rows, no copy
            h = ops.gemm(win, w, b, act=L.VX_ACT_GELU)
            T, c = h.shape
        return h

    def _positional(self, x):
        """x + GELU(grouped conv1d(x)) (Wav2Vec2PositionalConvEmbedding + Wav2Vec2SamePadLayer)."""
        P, cfg = self._P, self.cfg
        T, H = x.shape
        G, kp = cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings
        cg = H // G
        rows = T + kp                                       # kp // 2 zero rows in front, the rest behind
        xg = torch.zeros((G, rows, cg), device=x.device, dtype=ops.BF16)
        xg[:, kp // 2:kp // 2 + T].copy_(x.view(T, G, cg).transpose(0, 1))
        y = torch.empty_like(x)
        for g in range(G):
            win = torch.as_strided(xg, (T, kp * cg), (cg, 1), g * rows * cg)
            sl = slice(g * cg, (g + 1) * cg)
            ops.gemm(win, P.pos_w[g], P.pos_b[sl], act=L.VX_ACT_GELU, residual=x[:, sl], out=y[:, sl])
        return y

    def encode(self, wave):
        """One waveform, float32 [samples] on the device -> bf16 [T, hidden]."""
        P, cfg = self._prepared(), self.cfg
        eps = cfg.layer_norm_eps
        feats = self.extract_features(wave)
        x = ops.gemm(ops.layernorm(feats, P.fp_norm.g, P.fp_norm.b, eps), P.fp.w, P.fp.b)
        x = ops.layernorm(self._positional(x), P.enc_norm.g, P.enc_norm.b, eps)
        T = x.shape[0]
        for Lyr in P.layers:
            B._self_attention(Lyr.attn, x, x, seqs=1, n_tok=T, heads=cfg.num_attention_heads)   # x += attn(x)
            x = ops.layernorm(x, Lyr.ln1.g, Lyr.ln1.b, eps)
            mid = ops.gemm(x, Lyr.ff1.w, Lyr.ff1.b, act=L.VX_ACT_GELU)
            ops.gemm(mid, Lyr.ff2.w, Lyr.ff2.b, residual=x, out=x)
            x = ops.layernorm(x, Lyr.ln2.g, Lyr.ln2.b, eps)
        return x

    def forward(self, input_values, attention_mask=None, **kwargs):
        """input_values [B, samples] (or [samples]) -> namespace with `last_hidden_state` float32 [B, T, hidden]."""
        if attention_mask is not None:
            raise NotImplementedError("attention_mask: the V-Express pipeline never pads the waveform")
        wav = input_values.reshape(1, -1) if input_values.dim() == 1 else input_values
        wav = wav.to(device=self._device, dtype=torch.float32).contiguous()
        out = torch.stack([self.encode(wav[i]).float() for i in range(wav.shape[0])], dim=0)
        return SimpleNamespace(last_hidden_state=out)

    __call__ = forward
