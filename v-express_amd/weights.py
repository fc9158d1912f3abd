"""Weight ingestion: reference state_dict tensors -> device layouts the HIP kernels consume.

The wire format stays the reference's (`state_dict` keys of modules/unet_3d.py, unet_2d_condition.py,
diffusers AutoencoderKL — SURVEY.md Appendix C).  Re-layouts done once at load:
  * conv weights [Cout, Cin, kh, kw] -> [Cout, kh*kw*Cin] ((ky, kx, ci) K order, NHWC implicit GEMM), channel
    padding to multiples of 8 for the 4-channel latent convs;
  * attn1 / temporal to_q,to_k,to_v -> one fused [3C, C] matrix; cross-attention to_k,to_v -> [2C, ctx];
  * GEGLU `ff.net.0.proj` [8C, C]: value rows and gate rows interleaved in blocks of 8 so that one MFMA
    accumulator fragment pair holds (value, gate) of the same channels;
  * every ResnetBlock3D.time_emb_proj concatenated into one [sum(Cout), 1280] matrix (one GEMM per timestep);
  * weights -> bf16; biases, norm affine parameters and the motion-module sinusoid tables -> fp32.
"""
import torch

from . import lib as L



def __getattr__(name):
    # `weights.BF16` = the element dtype in force when a model prepares its device layouts (lib.element_type: torch.bfloat16,
    # or torch.float16 for a model in the reference's default --dtype fp16)
    if name == "BF16":
        return L.ELEM[0]
    raise AttributeError(name)


class Prepared(dict):
    """prefix -> SimpleNamespace-like dict of device tensors."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _dev(t, device, dtype):
    return t.detach().to(device=device, dtype=dtype).contiguous()


def pad8(n):
    return (n + 7) // 8 * 8


def prep_conv(sd, p, device, cin_pad=None, cout_pad=None):
    w = sd[p + ".weight"].detach().to(device=device, dtype=torch.float32)
    cout, cin, kh, kw = w.shape
    cin_p = cin_pad or pad8(cin)
    cout_p = cout_pad or pad8(cout)
    wp = torch.zeros((cout_p, kh, kw, cin_p), device=device, dtype=torch.float32)
    wp[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
    b = torch.zeros(cout_p, device=device, dtype=torch.float32)
    if (p + ".bias") in sd:
        b[:cout] = sd[p + ".bias"].detach().to(device=device, dtype=torch.float32)
    return Prepared(w=wp.reshape(cout_p, kh * kw * cin_p).to(L.ELEM[0]).contiguous(), b=b, k=kh, cout=cout)


def fold_upsample_phases(sd, p, device):
    """Nearest-2x upsampling + conv3x3 as four 2x2 convolutions over the original image (ops.upsample_conv_phases): output
    pixel (2y + a, 2x + b) reads input rows {y - 1 + a, y + a} and columns {x - 1 + b, x + b}; the 3x3 taps that land on the same
    input pixel are added in float32 and rounded once.  Rows: a = 0 -> {ky 0 | ky 1 + 2}, a = 1 -> {ky 0 + 1 | ky 2}; columns
    alike.  -> Prepared(w=[4 phases (a * 2 + b), Cout, 2 * 2 * Cin] elements, b=bias float32)."""
    w = sd[p + ".weight"].detach().to(device=device, dtype=torch.float32)          # [Cout, Cin, 3, 3]
    cout, cin, kh, kw = w.shape
    if (kh, kw) != (3, 3) or cin % 8 or cout % 8:
        return None
    rows = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
    phases = []
    for a in (0, 1):
        for b in (0, 1):
            wp = torch.zeros((cout, 2, 2, cin), device=device, dtype=torch.float32)
            for dy, kys in enumerate(rows[a]):
                for dx, kxs in enumerate(rows[b]):
                    for ky in kys:
                        for kx in kxs:
                            wp[:, dy, dx, :] += w[:, :, ky, kx]
            phases.append(wp.reshape(cout, 4 * cin))
    bias = torch.zeros(cout, device=device, dtype=torch.float32)
    if (p + ".bias") in sd:
        bias = sd[p + ".bias"].detach().to(device=device, dtype=torch.float32)
    return Prepared(w=torch.stack(phases).to(L.ELEM[0]).contiguous(), b=bias.contiguous())


def prep_linear(sd, p, device):
    w = _dev(sd[p + ".weight"], device, L.ELEM[0])
    if w.dim() == 4:     # 1x1 conv stored as [Cout, Cin, 1, 1]
        w = w.reshape(w.shape[0], w.shape[1]).contiguous()
    b = _dev(sd[p + ".bias"], device, torch.float32) if (p + ".bias") in sd else None
    return Prepared(w=w, b=b)


def prep_norm(sd, p, device):
    return Prepared(g=_dev(sd[p + ".weight"], device, torch.float32), b=_dev(sd[p + ".bias"], device, torch.float32))


LOG2E = 1.4426950408889634


def _cat_w(sd, keys, device, scales=None):
    """Row-concatenated bf16 weight; scales[i] (optional) multiplies tensor i in float32 BEFORE the one bf16 rounding."""
    parts = []
    for i, k in enumerate(keys):
        w = sd[k].detach().to(device=device, dtype=torch.float32)
        if scales is not None and scales[i] != 1.0:
            w = w * scales[i]
        parts.append(w.to(L.ELEM[0]))
    return torch.cat(parts, dim=0).contiguous()


def _cat_b(sd, keys, device, scales=None):
    if keys[0] not in sd:
        return None
    parts = [sd[k].detach().to(device=device, dtype=torch.float32) * (1.0 if scales is None else scales[i])
             for i, k in enumerate(keys)]
    return torch.cat(parts, dim=0).contiguous()


def key_fold(c, heads):
    """Factor folded into the to_k weights of an attention with `heads` heads over `c` channels: softmax scale
    d^-1/2 times log2(e), so that the kernels take q . k as the base-2 logit (vx_attention scale = 0) and neither
    operand is rounded to bf16 twice."""
    return (c // heads) ** -0.5 * LOG2E


def fold_layernorm(w_src, bias, gamma, beta, device, row_scale=None, interleave=False):
    """LayerNorm folded into the linear that consumes it (vx_gemm_params.ln_stats): from the fp32 source weight
    [N, K] (rows optionally scaled: the key fold), its bias (or None) and the norm's affine parameters build
      w = bf16(gamma (.) W)  (ONE rounding),  colsum[n] = sum_k float(w[n, k])  (of the rounded values the MFMA sees),
      b = bias + W beta  (fp32),
    so that LN(x) W^T + bias == rstd * (x w^T - mean * colsum) + b.  interleave: GEGLU row order for all three."""
    w32 = w_src.detach().to(device=device, dtype=torch.float32)
    if row_scale is not None:
        w32 = w32 * row_scale.to(device=device, dtype=torch.float32)[:, None]
    g = gamma.detach().to(device=device, dtype=torch.float32)
    bt = beta.detach().to(device=device, dtype=torch.float32)
    w = (w32 * g[None, :]).to(L.ELEM[0])
    colsum = w.float().sum(dim=1)
    b = w32 @ bt
    if bias is not None:
        b = b + bias.detach().to(device=device, dtype=torch.float32)
    if interleave:
        w, colsum, b = geglu_interleave(w), geglu_interleave(colsum), geglu_interleave(b)
    return Prepared(w=w.contiguous(), s=colsum.contiguous(), b=b.contiguous())


def fold_ff_proj(sd, ff_out, proj_out, device):
    """The block's proj_out folded into the feed-forward's second linear (round 6).  A transformer block ends with
        h' = h + W2 g + b2   (FeedForward.net.2 + residual),      out = Wp h' + bp + x_in   (proj_out + residual)
    and nothing else reads h' (modules/transformer_3d.py:150-169, modules/motion_module.py:172-182), so
        out = [h | g] [Wp | Wp W2]^T + (bp + Wp b2) + x_in
    is ONE dual-source GEMM of K = C + 4C instead of two launches with the [rows, C] tensor h' written and read back.
    w = bf16([Wp | Wp W2]) (the product in float32, ONE rounding), b = bp + Wp b2 (float32)."""
    w2 = sd[ff_out + ".weight"].detach().to(device=device, dtype=torch.float32)
    b2 = sd[ff_out + ".bias"].detach().to(device=device, dtype=torch.float32)
    wp = sd[proj_out + ".weight"].detach().to(device=device, dtype=torch.float32)
    if wp.dim() == 4:
        wp = wp.reshape(wp.shape[0], wp.shape[1])
    b = wp @ b2
    if (proj_out + ".bias") in sd:
        b = b + sd[proj_out + ".bias"].detach().to(device=device, dtype=torch.float32)
    w = torch.cat([wp, wp @ w2], dim=1).to(L.ELEM[0])
    return Prepared(w=w.contiguous(), b=b.contiguous())


def fold_groupnorm(w_src, bias, gamma, beta, device):
    """A GroupNorm WITHOUT activation folded into the 1x1 / linear layer that consumes it (ops.groupnorm_fold_linear):
    the frame-independent parts - the bf16 weight the per-frame copies are scaled from, gamma, and
    bb = bias + W beta (float32, from the float32 source weight)."""
    w32 = w_src.detach().to(device=device, dtype=torch.float32)
    if w32.dim() == 4:
        w32 = w32.reshape(w32.shape[0], w32.shape[1])
    bb = w32 @ beta.detach().to(device=device, dtype=torch.float32)
    if bias is not None:
        bb = bb + bias.detach().to(device=device, dtype=torch.float32)
    return Prepared(w=w32.to(L.ELEM[0]).contiguous(), g=gamma.detach().to(device=device, dtype=torch.float32).contiguous(),
                    bb=bb.contiguous())


def _qkv_rows(sd, p, device, heads, names):
    """fp32 row-concatenated source weight, bias (or None) and the per-row key-fold scale of a fused projection."""
    ws = [sd[f"{p}.{n}.weight"].detach().to(device=device, dtype=torch.float32) for n in names]
    scale = []
    for n, w in zip(names, ws):
        f = key_fold(w.shape[0], heads) if (heads is not None and n == "to_k") else 1.0
        scale.append(torch.full((w.shape[0],), f, device=device, dtype=torch.float32))
    bias = None
    if f"{p}.{names[0]}.bias" in sd:
        bias = torch.cat([sd[f"{p}.{n}.bias"].detach().to(device=device, dtype=torch.float32) * sc[0]
                          for n, sc in zip(names, scale)])
    return torch.cat(ws, dim=0), bias, torch.cat(scale)


def prep_self_attn(sd, p, device, heads=None):
    """diffusers Attention used as self-attention: fused QKV + out projection.  heads: fold the softmax scale into the
    key rows (`k_prescaled`)."""
    names = ["to_q", "to_k", "to_v"]
    scales = None
    if heads is not None:
        scales = [1.0, key_fold(sd[f"{p}.to_k.weight"].shape[0], heads), 1.0]
    return Prepared(wqkv=_cat_w(sd, [f"{p}.{n}.weight" for n in names], device, scales),
                    bqkv=_cat_b(sd, [f"{p}.{n}.bias" for n in names], device, scales),
                    out=prep_linear(sd, p + ".to_out.0", device), k_prescaled=heads is not None)


def prep_cross_attn(sd, p, device, heads=None):
    """Attention whose K/V come from another sequence: Q alone, fused KV, out projection.  heads: as prep_self_attn
    (the fused K | V weight only; `wk` stays unscaled)."""
    scales = None if heads is None else [key_fold(sd[p + ".to_k.weight"].shape[0], heads), 1.0]
    return Prepared(wq=_dev(sd[p + ".to_q.weight"], device, L.ELEM[0]),
                    wk=_dev(sd[p + ".to_k.weight"], device, L.ELEM[0]),
                    wkv=_cat_w(sd, [p + ".to_k.weight", p + ".to_v.weight"], device, scales),
                    out=prep_linear(sd, p + ".to_out.0", device), k_prescaled=heads is not None)


GEGLU_BLOCK = 8   # one 16-column MFMA fragment = 8 value columns + their 8 gate columns (vx_gemm GEGLU epilogue)


def geglu_interleave(t):
    """[8C, ...] (value rows then gate rows) -> blocks of 8 value rows followed by their 8 gate rows."""
    half = t.shape[0] // 2
    if half % 16:
        raise ValueError(f"GEGLU inner width {half} is not a multiple of 16")
    v = t[:half].reshape(half // GEGLU_BLOCK, GEGLU_BLOCK, *t.shape[1:])
    g = t[half:].reshape(half // GEGLU_BLOCK, GEGLU_BLOCK, *t.shape[1:])
    return torch.stack([v, g], dim=1).reshape(t.shape).contiguous()


def prep_ff(sd, p, device):
    w1 = sd[p + ".net.0.proj.weight"].detach().to(device=device, dtype=L.ELEM[0])
    b1 = sd[p + ".net.0.proj.bias"].detach().to(device=device, dtype=torch.float32)
    return Prepared(w1=geglu_interleave(w1), b1=geglu_interleave(b1), out=prep_linear(sd, p + ".net.2", device))


def prep_resnet(sd, p, device):
    r = Prepared(norm1=prep_norm(sd, p + ".norm1", device), conv1=prep_conv(sd, p + ".conv1", device),
                 norm2=prep_norm(sd, p + ".norm2", device), conv2=prep_conv(sd, p + ".conv2", device),
                 shortcut=None)
    if (p + ".conv_shortcut.weight") in sd:
        r["shortcut"] = prep_linear(sd, p + ".conv_shortcut", device)
    return r


def prep_spatial_read(sd, p, device, heads=None):
    """Transformer3DModel + TemporalBasicTransformerBlock (attn1, attn1_5, attn2, ff).  heads: fold the softmax scale
    into the key weights of the two flash-attention users (attn1, attn1_5)."""
    t = p + ".transformer_blocks.0"
    P = Prepared(norm=prep_norm(sd, p + ".norm", device), proj_in=prep_linear(sd, p + ".proj_in", device),
                 proj_out=prep_linear(sd, p + ".proj_out", device),
                 norm1=prep_norm(sd, t + ".norm1", device), attn1=prep_self_attn(sd, t + ".attn1", device, heads),
                 norm1_5=prep_norm(sd, t + ".norm1_5", device),
                 attn1_5=prep_cross_attn(sd, t + ".attn1_5", device, heads),
                 norm2=prep_norm(sd, t + ".norm2", device), attn2=prep_cross_attn(sd, t + ".attn2", device),
                 norm3=prep_norm(sd, t + ".norm3", device), ff=prep_ff(sd, t + ".ff", device))
    # the four LayerNorms folded into their consumer GEMMs (qkv, the two q projections, the GEGLU projection)
    w, b, rs = _qkv_rows(sd, t + ".attn1", device, heads, ["to_q", "to_k", "to_v"])
    P["ln_qkv"] = fold_layernorm(w, b, sd[t + ".norm1.weight"], sd[t + ".norm1.bias"], device, row_scale=rs)
    P["ln_q15"] = fold_layernorm(sd[t + ".attn1_5.to_q.weight"], None, sd[t + ".norm1_5.weight"],
                                 sd[t + ".norm1_5.bias"], device)
    P["ln_q2"] = fold_layernorm(sd[t + ".attn2.to_q.weight"], None, sd[t + ".norm2.weight"], sd[t + ".norm2.bias"],
                                device)
    P["ln_ff"] = fold_layernorm(sd[t + ".ff.net.0.proj.weight"], sd[t + ".ff.net.0.proj.bias"], sd[t + ".norm3.weight"],
                                sd[t + ".norm3.bias"], device, interleave=True)
    # the block's GroupNorm (no activation) folded into proj_in (used where ops.gn_fold_applies says so)
    P["gn_fold"] = fold_groupnorm(sd[p + ".proj_in.weight"], sd.get(p + ".proj_in.bias"), sd[p + ".norm.weight"],
                                  sd[p + ".norm.bias"], device)
    P["ff_proj"] = fold_ff_proj(sd, t + ".ff.net.2", p + ".proj_out", device)
    return P


def prep_spatial_write(sd, p, device, heads=None):
    """Transformer2DModel + BasicTransformerBlock of the ReferenceNet (attn1, attn2, ff)."""
    t = p + ".transformer_blocks.0"
    return Prepared(norm=prep_norm(sd, p + ".norm", device), proj_in=prep_linear(sd, p + ".proj_in", device),
                    proj_out=prep_linear(sd, p + ".proj_out", device),
                    norm1=prep_norm(sd, t + ".norm1", device), attn1=prep_self_attn(sd, t + ".attn1", device, heads),
                    norm2=prep_norm(sd, t + ".norm2", device), attn2=prep_cross_attn(sd, t + ".attn2", device),
                    norm3=prep_norm(sd, t + ".norm3", device), ff=prep_ff(sd, t + ".ff", device))


def prep_motion(sd, p, device):
    t = p + ".temporal_transformer"
    b = t + ".transformer_blocks.0"
    attn = []
    for i in range(2):
        a = f"{b}.attention_blocks.{i}"
        A = Prepared(attn=prep_self_attn(sd, a, device), norm=prep_norm(sd, f"{b}.norms.{i}", device),
                     pe=_dev(sd[a + ".pos_encoder.pe"][0], device, torch.float32))
        # LayerNorm folded into the fused QKV projection; the additive sinusoid table enters through the GEMM's
        # row-bias: (LN(x) + pe[f]) W^T = LN(x) W^T + pe[f] W^T, one float32 row per frame
        w, bq, _ = _qkv_rows(sd, a, device, None, ["to_q", "to_k", "to_v"])
        A["ln_qkv"] = fold_layernorm(w, bq, sd[f"{b}.norms.{i}.weight"], sd[f"{b}.norms.{i}.bias"], device)
        A["pe_rows"] = (A.pe @ w.to(L.ELEM[0]).float().t()).contiguous()              # [max_len, 3C]
        attn.append(A)
    P = Prepared(norm=prep_norm(sd, t + ".norm", device), proj_in=prep_linear(sd, t + ".proj_in", device),
                 proj_out=prep_linear(sd, t + ".proj_out", device), attn=attn,
                 ff_norm=prep_norm(sd, b + ".ff_norm", device), ff=prep_ff(sd, b + ".ff", device))
    P["ln_ff"] = fold_layernorm(sd[b + ".ff.net.0.proj.weight"], sd[b + ".ff.net.0.proj.bias"],
                                sd[b + ".ff_norm.weight"], sd[b + ".ff_norm.bias"], device, interleave=True)
    P["gn_fold"] = fold_groupnorm(sd[t + ".proj_in.weight"], sd.get(t + ".proj_in.bias"), sd[t + ".norm.weight"],
                                  sd[t + ".norm.bias"], device)
    P["ff_proj"] = fold_ff_proj(sd, b + ".ff.net.2", t + ".proj_out", device)
    return P
