"""Once-per-clip prologue models on libvexpress_hip kernels: VKpsGuider and AudioProjection
with the reference's constructor / `load_state_dict` / `forward` surface, plus the audio-window construction of
`VExpressPipeline.prepare_audio_embeddings`.

Both models are compositions of the hot-path kernels: every convolution is the implicit-GEMM `vx_gemm` (3x3, stride
1 / 2, SiLU fused in the epilogue), the Perceiver blocks are `vx_layernorm` + `vx_gemm` + `vx_small_kv_attention`
(15 keys) with the erf-GELU fused into the first feed-forward GEMM.  Activations are bf16 channels-last tokens like
the rest of the path; the kps features are produced directly in the `[b, F, hw, 320]` token layout the denoising
loop consumes (the reference moves them to the CPU and back every window, pipelines/v_express_pipeline.py:363,531).
"""

import torch

from . import lib as L
from . import ops
from . import weights as Wt
from .module_base import DeviceModule
from .synth import AudioProjectionConfig, KpsGuiderConfig


class _Module(DeviceModule):
    """Prologue models load strictly like the reference does (inference.py:101,127): `_schema()` names the synth
    generator of the reference module's state_dict, evaluated on the meta device for names and shapes."""

    def _schema(self):
        return None

    def expected_keys(self):
        if getattr(self, "_expected", None) is None:
            sd = self._schema()
            self._expected = None if sd is None else {k: tuple(v.shape) for k, v in sd.items()}
        return self._expected


class VKpsGuider(_Module):
    """modules/v_kps_guider.py:10-45.  `forward(conditioning [b, 3, f, H, W])` -> `[b, C, f, H/8, W/8]` float32 like
    the reference; `forward_tokens` returns the bf16 token layout `[b*f, (H/8)*(W/8), C]` without the transposes."""

    def __init__(self, conditioning_embedding_channels=320, conditioning_channels=3,
                 block_out_channels=(16, 32, 96, 256)):
        super().__init__()
        self.cfg = KpsGuiderConfig(conditioning_embedding_channels, conditioning_channels, tuple(block_out_channels))

    def _schema(self):
        from . import synth
        return synth.kps_guider_state_dict(self.cfg, device="meta")

    def _prepared(self):
        if self._P is None:
            self._need_gpu()
            sd, dev = self._raw, self._device
            n = 2 * (len(self.cfg.block_out_channels) - 1)
            self._P = Wt.Prepared(conv_in=Wt.prep_conv(sd, "conv_in", dev),
                                  blocks=[Wt.prep_conv(sd, f"blocks.{i}", dev) for i in range(n)],
                                  conv_out=Wt.prep_conv(sd, "conv_out", dev))
        return self._P

    def forward_tokens(self, conditioning):
        P = self._prepared()
        b, c, f, H, W = conditioning.shape
        n = b * f
        x = ops.ncfhw_to_nhwc(conditioning.to(self._device).float().contiguous(), 8).view(n * H * W, 8)
        x = ops.gemm(x, P.conv_in.w, P.conv_in.b, geom=ops.ConvGeom(n, H, W, 3, 3, 1, 1), act=L.VX_ACT_SILU)
        for i, blk in enumerate(P.blocks):
            g = ops.ConvGeom(n, H, W, 3, 3, 2 if i % 2 else 1, 1)
            x = ops.gemm(x, blk.w, blk.b, geom=g, act=L.VX_ACT_SILU)
            H, W = g.h_out, g.w_out
        x = ops.gemm(x, P.conv_out.w, P.conv_out.b, geom=ops.ConvGeom(n, H, W, 3, 3, 1, 1))
        return x.view(n, H * W, -1), H, W

    def forward(self, conditioning):
        b, _, f, _, _ = conditioning.shape
        tok, h, w = self.forward_tokens(conditioning)
        c = tok.shape[-1]
        return tok.float().view(b, f, h, w, c).permute(0, 4, 1, 2, 3).contiguous()

    __call__ = forward


class AudioProjection(_Module):
    """modules/audio_projection.py:97-150 (num_latents_mean_pooled = 0, the only configuration V-Express uses)."""

    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, max_seq_len=257, num_latents_mean_pooled=0):
        super().__init__()
        if num_latents_mean_pooled:
            raise NotImplementedError("num_latents_mean_pooled > 0 is not used by V-Express (inference.py:116-126)")
        self.cfg = AudioProjectionConfig(dim, depth, dim_head, heads, num_queries, embedding_dim, output_dim, ff_mult,
                                         max_seq_len)

    def _schema(self):
        from . import synth
        return synth.audio_projection_state_dict(self.cfg, device="meta")

    def _prepared(self):
        if self._P is None:
            self._need_gpu()
            sd, dev, cfg = self._raw, self._device, self.cfg
            f32 = dict(device=dev, dtype=torch.float32)
            P = Wt.Prepared(pos=sd["pos_emb.weight"].to(**f32), latents=sd["latents"].to(**f32)[0],
                            proj_in=Wt.prep_linear(sd, "proj_in", dev), proj_out=Wt.prep_linear(sd, "proj_out", dev),
                            norm_out=Wt.prep_norm(sd, "norm_out", dev), layers=[])
            for i in range(cfg.depth):
                a, f = f"layers.{i}.0", f"layers.{i}.1"
                P.layers.append(Wt.Prepared(
                    norm1=Wt.prep_norm(sd, a + ".norm1", dev), norm2=Wt.prep_norm(sd, a + ".norm2", dev),
                    wq=Wt._dev(sd[a + ".to_q.weight"], dev, Wt.BF16), wkv=Wt._dev(sd[a + ".to_kv.weight"], dev, Wt.BF16),
                    wo=Wt._dev(sd[a + ".to_out.weight"], dev, Wt.BF16), ff_norm=Wt.prep_norm(sd, f + ".0", dev),
                    w1=Wt._dev(sd[f + ".1.weight"], dev, Wt.BF16), w2=Wt._dev(sd[f + ".3.weight"], dev, Wt.BF16)))
            self._P = P
        return self._P

    def forward(self, x):
        """x [F, n, embedding_dim] -> [F, num_queries, output_dim] float32."""
        P, cfg = self._prepared(), self.cfg
        Fn, n, _ = x.shape
        nq, dim, inner = cfg.num_queries, cfg.dim, cfg.dim_head * cfg.heads
        n_kv = n + nq
        x = (x.to(self._device).float() + P.pos[:n]).to(ops.BF16).reshape(Fn * n, -1).contiguous()   # :131-134
        xt = ops.gemm(x, P.proj_in.w, P.proj_in.b)                                                   # :138
        lat = P.latents.to(ops.BF16).repeat(Fn, 1).contiguous()                                      # :136  [F*nq, dim]
        kvin = torch.empty((Fn, n_kv, dim), device=self._device, dtype=ops.BF16)
        for Lyr in P.layers:
            # PerceiverAttention (:48-85): keys/values = LN1(x) ++ LN2(latents), queries = LN2(latents)
            xn = ops.layernorm(xt, Lyr.norm1.g, Lyr.norm1.b)
            ln = ops.layernorm(lat, Lyr.norm2.g, Lyr.norm2.b)
            kvin[:, :n].copy_(xn.view(Fn, n, dim))
            kvin[:, n:].copy_(ln.view(Fn, nq, dim))
            q = ops.gemm(ln, Lyr.wq)
            kv = ops.gemm(kvin.view(Fn * n_kv, dim), Lyr.wkv)
            a = ops.small_kv_attention(q, kv, batch=Fn, n_q=nq, n_kv=n_kv, heads=cfg.heads, head_dim=cfg.dim_head)
            ops.gemm(a, Lyr.wo, residual=lat, out=lat)                                               # :145
            # FeedForward (:88-95): LN, Linear, GELU (erf), Linear, + residual
            h = ops.gemm(ops.layernorm(lat, Lyr.ff_norm.g, Lyr.ff_norm.b), Lyr.w1, act=L.VX_ACT_GELU)
            ops.gemm(h, Lyr.w2, residual=lat, out=lat)                                               # :146
        out = ops.gemm(lat, P.proj_out.w, P.proj_out.b)
        out = ops.layernorm(out, P.norm_out.g, P.norm_out.b)
        return out.float().view(Fn, nq, -1)

    __call__ = forward


def audio_windows(last_hidden_state, video_length, num_pad_audio_frames):
    """VExpressPipeline.prepare_audio_embeddings, pipelines/v_express_pipeline.py:381-401 (data movement only):
    wav2vec2 states [1, T, d] -> linear interpolation to 2*video_length rows, 2*num_pad zero rows on both sides, one
    window of 2*(2*num_pad+1) rows per frame -> [video_length, 2*(2*num_pad+1), d]."""
    emb = torch.nn.functional.interpolate(last_hidden_state.float().permute(0, 2, 1), size=2 * video_length,
                                          mode="linear")[0].permute(1, 0)
    pad = torch.zeros_like(emb)[:2 * num_pad_audio_frames]
    emb = torch.cat([pad, emb, pad], dim=0)
    idx = (2 * torch.arange(video_length, device=emb.device)[:, None] +
           torch.arange(2 * (2 * num_pad_audio_frames + 1), device=emb.device)[None])
    return emb[idx]
