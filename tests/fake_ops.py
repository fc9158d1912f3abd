"""Torch/CPU stand-ins for a few `v_express_amd.ops` entry points — TEST INFRASTRUCTURE for the CPU suite only.

They let the *host composition* of a model (weight re-layouts, strided window views, buffer plumbing, call order) be
checked against the oracle in the dev container, where no GPU exists.  Same signatures and output dtypes (bf16 rounding
at every kernel boundary) as the real wrappers; the arithmetic inside is float64 torch, so - like the real kernels - a
row's result does not depend on how many other rows share the call (fp32 BLAS blocking would).  Never imported by the
package; the GPU tests run the same model code on the real kernels.
"""
import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


def _act(y, act):
    return F.silu(y) if act == 1 else F.gelu(y) if act == 2 else y


def wave_conv1d(wave, wt, stride):
    taps, _ = wt.shape
    return (wave.double().unfold(0, taps, stride) @ wt.double()).to(BF16)


def _gn_ws(x, frames, hw, groups):
    """Emulated workspace: (mean, variance) per (frame, group) in float64 - opaque to the host code."""
    xg = x.double().reshape(frames, hw, groups, -1)
    return torch.stack([xg.mean(dim=(1, 3)), xg.var(dim=(1, 3), unbiased=False)], dim=-1)      # [frames, groups, 2]


def groupnorm(x1, gamma, beta, *, frames, hw, groups, eps, silu, x2=None, out=None, pad_hw=None):
    from v_express_amd import ops as real_ops
    st = real_ops.gn_of(x1) if x2 is None else None
    if st is not None:
        # statistics the producing GEMM attached (vx_gemm_params.gn_ws): they must describe THIS tensor as it is now
        assert st.fits(frames, hw, groups, x1.shape[-1]), "stale GroupNorm statistics: geometry"
        assert torch.equal(st.ws, _gn_ws(x1, frames, hw, groups)), "stale GroupNorm statistics: values"
    x = x1.double().view(frames, hw, -1)
    if x2 is not None:
        x = torch.cat([x, x2.double().view(frames, hw, -1)], dim=-1)
    c = x.shape[-1]
    y = F.group_norm(x.transpose(1, 2), groups, gamma.double(), beta.double(), eps) if hw * (c // groups) > 1 else \
        beta.double().expand(frames, c)[:, :, None].expand(frames, c, hw)
    y = _act(y, int(silu)).transpose(1, 2).to(BF16)
    if pad_hw is not None:
        H, W = pad_hw
        assert out is None and H * W == hw
        buf = real_ops.padded_buffer(x1.device, frames, H, W, c)
        buf.view(frames, H + 2, W + 2, c)[:, 1:H + 1, 1:W + 1].copy_(y.reshape(frames, H, W, c))
        return buf
    if out is not None:
        out.copy_(y.reshape(out.shape))
        return out
    return y.contiguous()


def groupnorm_stats(x1, *, frames, hw, groups, x2=None):
    from v_express_amd import ops as real_ops
    st = real_ops.gn_of(x1) if x2 is None else None
    if st is not None and st.fits(frames, hw, groups, x1.shape[-1]):
        assert torch.equal(st.ws, _gn_ws(x1, frames, hw, groups)), "stale GroupNorm statistics: values"
        return st.ws, st.slabs
    x = x1.double().view(frames, hw, -1)
    if x2 is not None:
        x = torch.cat([x, x2.double().view(frames, hw, -1)], dim=-1)
    return _gn_ws(x, frames, hw, groups), 1


def groupnorm_fold_linear(ws, gamma, w, bias_beta, *, frames, hw, groups, eps, slices=None):
    n, c = w.shape
    mean = ws[..., 0].repeat_interleave(c // groups, dim=1)                                    # [frames, c]
    rstd = torch.rsqrt(ws[..., 1] + eps).repeat_interleave(c // groups, dim=1)
    w_f = (w.double()[None] * (gamma.double()[None] * rstd)[:, None, :]).to(BF16)              # one rounding
    b_f = bias_beta.double()[None] - (w_f.double() * mean[:, None, :]).sum(dim=-1)
    return w_f.contiguous(), b_f.float().contiguous()


def layernorm(x, gamma, beta, eps=1e-5, *, add=None, add_rows_per_entry=1, add_entries=1, out=None):
    assert x.dtype == BF16
    x2 = x.reshape(-1, x.shape[-1]) if x.is_contiguous() else x
    y = F.layer_norm(x2.double(), (x2.shape[-1],), gamma.double(), beta.double(), eps)
    if add is not None:
        idx = (torch.arange(y.shape[0]) // add_rows_per_entry) % add_entries
        y = y + add.double().reshape(-1, y.shape[-1])[idx]
    y = y.to(BF16)
    if out is not None:
        out.copy_(y)
        return out
    return y


def row_stats(x, eps=1e-5, out=None):
    """vx_row_stats: (mean, rstd) per row; into a [rows, 4] buffer (vx_row_stats_parts): (sum, sum of squares) of each half."""
    x2 = (x.reshape(-1, x.shape[-1]) if x.is_contiguous() else x).double()
    if out is not None and out.shape[1] == 4:
        h = x2.shape[1] // 2
        st = torch.stack([x2[:, :h].sum(1), (x2[:, :h] ** 2).sum(1), x2[:, h:].sum(1), (x2[:, h:] ** 2).sum(1)], dim=1).float()
    else:
        mean = x2.mean(dim=1)
        var = x2.var(dim=1, unbiased=False)
        st = torch.stack([mean, torch.rsqrt(var + eps)], dim=1).float()
    if out is not None:
        assert out.dtype == torch.float32 and out.is_contiguous() and out.shape == st.shape
        out.copy_(st)
        return out
    return st


def _apply_ln(y, ln, eps=1e-5):
    """vx_gemm_params.ln_stats: acc <- rstd[m] * (acc - mean[m] * colsum[n]); [m, 4] statistics = two-part sums over the k
    columns of the rows (ln_stats_parts = 2), finished here like the epilogue does"""
    if ln is None:
        return y
    stats, colsum, *k = ln
    st = stats.double()
    if stats.shape[1] == 4:
        kk = k[0]
        mean = (st[:, 0] + st[:, 2]) / kk
        var = ((st[:, 1] + st[:, 3]) / kk - mean * mean).clamp_min(0)
        st = torch.stack([mean, torch.rsqrt(var + eps)], dim=1)
    return st[:, 1:2] * (y - st[:, 0:1] * colsum.double()[None, :])


def layernorm_fp8(x, gamma, beta, eps=1e-5, *, add=None, add_rows_per_entry=1, add_entries=1):
    """vx_layernorm_fp8: (optional) LayerNorm in float64, per-row scale = max|y| / 448, OCP e4m3 round-to-nearest."""
    from v_express_amd import ops as real_ops
    x2 = x.reshape(-1, x.shape[-1]) if x.is_contiguous() else x
    y = x2.double()
    if gamma is not None:
        y = F.layer_norm(y, (y.shape[-1],), gamma.double(), beta.double(), eps)
        if add is not None:
            idx = (torch.arange(y.shape[0]) // add_rows_per_entry) % add_entries
            y = y + add.double().reshape(-1, y.shape[-1])[idx]
    y = y.float()
    amax = y.abs().amax(dim=1)
    sc = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    k = y.shape[1]
    q = torch.zeros((y.shape[0], real_ops.pad128(k)), dtype=torch.uint8)
    q[:, :k] = (y / sc[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    return real_ops.Fp8Rows(q, sc, k)


def quantize_fp8(x):
    return layernorm_fp8(x, None, None)


def _dequant(a, w):
    """(Fp8Rows, Fp8Weight) -> float64 operands (the fp8 MFMA is exact on e4m3 products, fp32 accumulation)."""
    x = a.q.view(torch.float8_e4m3fn).double() * a.scale.double()[:, None]
    wt = w.w8.view(torch.float8_e4m3fn).double() * w.scale.double()[:, None]
    return x, wt


def _is_fp8(a):
    return type(a).__name__ == "Fp8Rows"


def _conv_rows(a, a2, w, geom):
    """Implicit-GEMM semantics of vx_gemm: rows (frame, y, x) of NHWC sources (channel-concatenated), weight [N, kh*kw*C]
    with (ky, kx, c) column order, nearest-2x upsample before the conv, zero padding `pad` before each spatial axis and
    whatever the output size needs after it."""
    x = a.double() if a2 is None else torch.cat([a.double(), a2.double()], dim=-1)
    nb, h, wd = geom.nb, geom.h_in, geom.w_in
    c = x.shape[-1]
    n = w.shape[0]
    assert x.shape[0] == nb * h * wd and w.shape[1] == geom.kh * geom.kw * c, (x.shape, w.shape, vars(geom))
    img = x.view(nb, h, wd, c).permute(0, 3, 1, 2)
    if geom.upsample:
        img = F.interpolate(img, scale_factor=2, mode="nearest")
    he, we = img.shape[-2:]
    need_h = (geom.h_out - 1) * geom.stride + geom.kh - he
    need_w = (geom.w_out - 1) * geom.stride + geom.kw - we
    img = F.pad(img, (geom.pad, max(need_w - geom.pad, 0), geom.pad, max(need_h - geom.pad, 0)))
    wt = w.double().view(n, geom.kh, geom.kw, c).permute(0, 3, 1, 2)
    y = F.conv2d(img, wt, stride=geom.stride)[:, :, :geom.h_out, :geom.w_out]
    return y.permute(0, 2, 3, 1).reshape(geom.m, n)


def gemm(a, w, bias=None, *, geom=None, a2=None, residual=None, alpha=1.0, act=0, rowbias=None, rows_per_group=0,
         out=None, out_f32=False, ln=None, stats_out=None, stats_eps=1e-5, w_group_rows=0, gn=None):
    if w_group_rows:                                   # per-row-group weights [groups, N, K] (vx_gemm_params.w_group_rows)
        assert geom is None and a2 is None and w.dim() == 3 and w.is_contiguous() and a.dtype == BF16
        assert a.shape[0] == w.shape[0] * w_group_rows
        y = torch.einsum("grk,gnk->grn", a.double().view(w.shape[0], w_group_rows, -1), w.double())
        y = y.reshape(a.shape[0], w.shape[1])
    elif _is_fp8(a):
        assert geom is None and a2 is None
        x, wt = _dequant(a, w)
        y = x @ wt.t()
    elif geom is None:
        assert a.dtype == BF16 and w.dtype == BF16 and a.stride(-1) == 1 and w.is_contiguous()
        assert a.dim() == 2 and a.stride(0) % 8 == 0
        x = a.double() if a2 is None else torch.cat([a.double(), a2.double()], dim=-1)
        assert x.shape[1] == w.shape[1], (x.shape, w.shape)
        y = x @ w.double().t()
    else:
        assert a.dtype == BF16 and w.dtype == BF16 and a.stride(-1) == 1 and w.is_contiguous()
        y = _conv_rows(a.reshape(-1, a.shape[-1]), None if a2 is None else a2.reshape(-1, a2.shape[-1]), w, geom)
    y = _apply_ln(y, None if ln is None else (*ln, a.shape[-1]))
    if bias is not None:
        assert bias.dtype == torch.float32
        y = y + bias
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rows_per_group > 0
        grp = torch.arange(y.shape[0]) // rows_per_group
        y = y + rowbias[grp, :y.shape[1]]
    y = _act(y, act)
    # the accumulator + bias is a float32 value in the kernel; the residual is added to it in float32, then ONE rounding to
    # the element type (so that add_residual_f32(x, gemm(..., out_f32=True)) has the bits of gemm(..., residual=x))
    y = y.float()
    if residual is not None:
        y = residual.float().reshape(y.shape) + (y if alpha == 1.0 else torch.tensor(alpha, dtype=torch.float32) * y)
    elif alpha != 1.0:
        y = torch.tensor(alpha, dtype=torch.float32) * y
    y = y if out_f32 else y.to(BF16)
    if stats_out is not None:                          # statistics of the STORED rows (vx_gemm_params.row_stats_out)
        assert not out_f32
        row_stats(y, stats_eps, out=stats_out)
    if out is not None:
        assert out.shape[-1] == y.shape[-1] and out.stride(-1) == 1
        out.reshape(y.shape).copy_(y) if out.is_contiguous() else out.copy_(y)
        y = out
    if gn is not None and not out_f32 and not _is_fp8(a):
        # GroupNorm partial sums of the STORED values ride on the tensor object (vx_gemm_params.gn_ws); the emulation
        # produces them whenever asked (the kernels only where vx_gemm_gn_slabs allows), one slab per frame
        from v_express_amd import ops as real_ops
        groups, hw = gn
        if real_ops.GN_FUSED[0] and y.shape[0] % hw == 0 and y.shape[1] % groups == 0:
            frames = y.shape[0] // hw
            y._vx_gn = real_ops.GnStats(_gn_ws(y, frames, hw, groups), 1, groups, frames, hw, y.shape[1])
    return y


def geglu(a, w_interleaved, bias_interleaved, out=None, ln=None):
    y = _apply_ln(a.double() @ w_interleaved.double().t(), None if ln is None else (*ln, a.shape[-1]))
    if bias_interleaved is not None:
        y = y + bias_interleaved
    blk = y.view(y.shape[0], -1, 2, 8)                      # blocks of 8 value columns followed by their 8 gates
    r = (blk[:, :, 0] * F.gelu(blk[:, :, 1])).reshape(y.shape[0], -1).to(BF16)
    if out is not None:
        out.copy_(r)
        return out
    return r


def ff_fused(h, w1_folded, b1, colsum, stats, w2, b2):
    """vx_ff_fused = the two launches it replaces, in place on h (the kernel is bit-identical to them on the GPU)."""
    g = geglu(h, w1_folded, b1, ln=(stats, colsum))
    return gemm(g, w2, b2, residual=h, out=h)


def tblock_fused(h, wqkv_folded, bqkv, colsum, pe_rows, wo, bo, *, b, f, hw, heads, stats=None, stats_out=None, eps=1e-5):
    """vx_tblock_fused = the three launches it replaces, in place on h (statistics: the rows' own when none are given)."""
    if stats is None:
        stats = row_stats(h, eps)
    m, c = h.shape
    rb = pe_rows[:f].repeat(b, 1).contiguous() if pe_rows is not None else None
    qkv = gemm(h, wqkv_folded, bqkv, rowbias=rb, rows_per_group=hw, ln=(stats, colsum))
    a = temporal_attention(qkv, b=b, f=f, hw=hw, heads=heads, head_dim=c // heads)
    return gemm(a, wo, bo, residual=h, out=h, stats_out=stats_out, stats_eps=eps)


def alloc_vt(seqs, heads, head_dim, n, device):
    return torch.zeros((seqs, heads, head_dim, (n + 7) // 8 * 8), device=device, dtype=BF16)


def gemm_split(a, w, bias, parts, *, part_cols, seq_len=0, head_dim=0, geom=None, ln=None):
    if _is_fp8(a):
        x, wt = _dequant(a, w)
        y = x @ wt.t()
    else:
        y = _apply_ln(a.double() @ w.double().t(), None if ln is None else (*ln, a.shape[-1]))
    if bias is not None:
        y = y + bias
    for i, (kind, t) in enumerate(parts):
        col = y[:, i * part_cols:(i + 1) * part_cols]
        if kind == "rows":
            t.copy_(col.to(BF16))
        else:
            seqs, heads = t.shape[0], t.shape[1]
            t[..., :seq_len] = col.view(seqs, seq_len, heads, head_dim).permute(0, 2, 3, 1).to(BF16)


def key_norm_max(k, *, kv_batches, heads, n_kv, head_dim):
    return k.double().reshape(kv_batches, n_kv, -1)[..., :heads * head_dim].reshape(
        kv_batches, n_kv, heads, head_dim).norm(dim=-1).amax(dim=1).reshape(-1).float()


def attention(q, k, vt, *, batch, heads, n_q, n_kv, head_dim, q_per_kv=1, out=None, kmax=None, k_prescaled=False):
    c = heads * head_dim
    kvb = batch // q_per_kv
    qh = q.double().reshape(batch, n_q, heads, head_dim).transpose(1, 2)
    kh = k.double().reshape(kvb, n_kv, heads, head_dim).transpose(1, 2).repeat_interleave(q_per_kv, dim=0)
    if k_prescaled:      # k carries d^-1/2 log2(e): softmax_j 2^(q.k_j)  ==  SDPA on k * sqrt(d) * ln 2
        kh = kh * (head_dim ** 0.5 * 0.6931471805599453)
    vh = vt[..., :n_kv].double().transpose(-1, -2).repeat_interleave(q_per_kv, dim=0)
    o = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(batch * n_q, c).to(BF16)
    if out is not None:
        out.copy_(o)
        return out
    return o


def temporal_attention(qkv, *, b, f, hw, heads, head_dim, out=None):
    c = heads * head_dim
    q, k, v = (t.reshape(b, f, hw, heads, head_dim).permute(0, 2, 3, 1, 4) for t in qkv.double().chunk(3, dim=-1))
    o = F.scaled_dot_product_attention(q, k, v).permute(0, 3, 1, 2, 4).reshape(b * f * hw, c).to(BF16)
    if out is not None:
        out.copy_(o)
        return out
    return o


def small_kv_attention(q, kv, *, batch, n_q, n_kv, heads, head_dim, out=None):
    c = heads * head_dim
    qh = q.double().reshape(batch, n_q, heads, head_dim).transpose(1, 2)
    kvf = kv.double().reshape(batch, n_kv, 2 * c)
    kh = kvf[..., :c].reshape(batch, n_kv, heads, head_dim).transpose(1, 2)
    vh = kvf[..., c:].reshape(batch, n_kv, heads, head_dim).transpose(1, 2)
    o = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(batch * n_q, c).to(BF16)
    if out is not None:
        out.copy_(o)
        return out
    return o


class _FakeAudioFold:
    def __init__(self, kq, colsum, sbias, vo, frames, c, heads, n_ctx):
        self.kq, self.colsum, self.sbias, self.vo, self.frames, self.c, self.heads, self.n_ctx = kq, colsum, sbias, vo, frames, c, heads, n_ctx


def audio_xattn_pack(kv, wq_folded, bq_folded, wo, *, frames, n_ctx, heads):
    """vx_audio_xattn_pack with the kernel's rounding points: Kq_f[(h, t), :] = bf16(log2(e) / sqrt(d) K_f,h[t] Wq_h),
    colsum of the ROUNDED rows, sbias in float, VO_f[:, (h, t)] = bf16(Wo_h V_f,h[t])  (natural [frames, heads, n_ctx, C] order)."""
    c = wo.shape[0]
    d = c // heads
    scale = 1.4426950408889634 / d ** 0.5
    k = kv[:, :c].double().reshape(frames, n_ctx, heads, d)
    v = kv[:, c:].double().reshape(frames, n_ctx, heads, d)
    wq = wq_folded.double().reshape(heads, d, c)                        # rows h d + j
    kq = (torch.einsum("fthj,hjc->fhtc", k, wq) * scale).to(BF16)
    colsum = kq.double().sum(-1)
    sbias = torch.zeros(frames, heads, n_ctx, dtype=torch.float64)
    if bq_folded is not None:
        sbias = torch.einsum("fthj,hj->fht", k, bq_folded.double().reshape(heads, d)) * scale
    wo_h = wo.double().reshape(c, heads, d)
    vo = torch.einsum("nhj,fthj->fhtn", wo_h, v).to(BF16)
    return _FakeAudioFold(kq, colsum, sbias, vo, frames, c, heads, n_ctx)


def audio_xattn(h, stats, fold, bias_o, alpha, *, rows_per_frame, stats_out=None, stats_eps=1e-5, out=None):
    """vx_audio_xattn: S = LN-folded x Kq^T (+ sbias), base-2 softmax per head, P rounded to bf16, y = x + alpha (P VO^T + b)."""
    m, c = h.shape
    assert m == fold.frames * rows_per_frame and c == fold.c
    x = h.double().reshape(fold.frames, rows_per_frame, c)
    st = stats.double()
    if stats.shape[1] == 4:
        mean = (st[:, 0] + st[:, 2]) / c
        var = ((st[:, 1] + st[:, 3]) / c - mean * mean).clamp_min(0)
        st = torch.stack([mean, torch.rsqrt(var + 1e-5)], dim=1)
    mean = st[:, 0].reshape(fold.frames, rows_per_frame, 1, 1)
    rstd = st[:, 1].reshape(fold.frames, rows_per_frame, 1, 1)
    s = torch.einsum("frc,fhtc->frht", x, fold.kq.double())
    s = rstd * (s - mean * fold.colsum[:, None]) + fold.sbias[:, None]
    p = torch.softmax(s * 0.6931471805599453, dim=-1).to(BF16).double()        # 2^s = e^(s ln 2)
    y = torch.einsum("frht,fhtn->frn", p, fold.vo.double()) + bias_o.double()
    y = (x + alpha * y).reshape(m, c).to(BF16)
    if stats_out is not None:
        row_stats(y, stats_eps, out=stats_out)
    dst = h if out is None else out
    dst.copy_(y)
    return dst


def upsample_conv_phases(x, w_phases, bias, *, frames, H, W):
    """ops.upsample_conv_phases: four pad-0 2x2 convolutions over the zero-bordered image (float64 accumulate, + bias, one
    rounding per phase output) interleaved into the 2H x 2W image."""
    cin, cout = x.shape[-1], w_phases.shape[1]
    xp = F.pad(x.double().reshape(frames, H, W, cin).permute(0, 3, 1, 2), (1, 1, 1, 1))
    out = torch.empty((frames, 2 * H, 2 * W, cout), dtype=BF16)
    for a in (0, 1):
        for b in (0, 1):
            w = w_phases[a * 2 + b].double().reshape(cout, 2, 2, cin).permute(0, 3, 1, 2)
            y = F.conv2d(xp[:, :, a:a + H + 1, b:b + W + 1], w) + bias.double()[None, :, None, None]
            out[:, a::2, b::2] = y.float().to(BF16).permute(0, 2, 3, 1)
    return out.reshape(frames, 4 * H * W, cout)


def add_residual_f32(x, y32, out=None):
    """vx_add_residual_f32: the residual add (float32) and the one rounding of the STORE epilogue."""
    assert y32.dtype == torch.float32 and y32.shape == x.shape
    r = (x.float() + y32).to(BF16)
    if out is not None:
        out.copy_(r)
        return out
    return r


def add_row_bias(x, bias, alpha=1.0):
    x.copy_((x.double() + alpha * bias).to(BF16))
    return x


def gather_latents(latents, frame_ids, reps, c_pad=8):
    _, c, _, h, w = latents.shape
    x = latents[0][:, frame_ids.long()].reshape(c, -1, h * w).permute(1, 2, 0)            # [f, hw, C]
    out = torch.zeros((reps, x.shape[0], h * w, c_pad), dtype=BF16)
    out[..., :c] = x.to(BF16)
    return out.reshape(reps * x.shape[0], h * w, c_pad)


def cfg_combine(unet_out, c, f, hw, guidance, pred_slot):
    u, cnd = unet_out[:f * hw, :c].double(), unet_out[f * hw:2 * f * hw, :c].double()
    pred_slot.copy_((u + guidance * (cnd - u)).reshape(f, hw, c).permute(2, 0, 1))


def pack_rows(src, c, dst):
    dst.view(-1, c).copy_(src[:, :c])


def combine_units(gathered, unit_index, c, f, hw, guidance, preds):
    nW, halves, S = unit_index.shape
    f_loc = f // S
    g = gathered.reshape(-1, f_loc * hw, c)
    for wi in range(nW):
        h = [torch.cat([g[int(unit_index[wi, hh, j])] for j in range(S)], dim=0) for hh in range(halves)]   # [f*hw, c]
        u, cnd = h[0].double(), h[-1].double()
        r = u + guidance * (cnd - u)
        preds[wi].copy_(r.view(f, hw, c).permute(2, 0, 1))


def overlap_ddim_step(latents, preds, terms, frame_ids, counts, coef):
    sa, s1a, sap, s1ap = (float(v) for v in coef)
    _, c, _, h, w = latents.shape
    new = {}
    for i, fr in enumerate(frame_ids.tolist()):
        v = None
        for slot, li in terms[i].tolist():
            if slot < 0:
                continue
            term = preds[slot, :, li] / counts[i]
            v = term if v is None else v + term
        x = latents[0, :, fr].reshape(c, h * w)
        new[fr] = (sap * (sa * x - s1a * v) + s1ap * (sa * v + s1a * x)).reshape(c, h, w)
    for fr, val in new.items():
        latents[0, :, fr] = val


def ncfhw_to_nhwc(x, c_pad=None):
    b, c, f, h, w = x.shape
    c_pad = c_pad or (c + 7) // 8 * 8
    out = torch.zeros((b * f, h * w, c_pad), dtype=BF16)
    out[..., :c] = x.double().permute(0, 2, 3, 4, 1).reshape(b * f, h * w, c).to(BF16)
    return out


def nhwc_to_ncfhw(x, b, c, f, h, w):
    return x[:, :c].float().reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3).contiguous()


def vae_postprocess(x, n, c, h, w):
    return (x[:, :c].float().reshape(n, h, w, c).permute(0, 3, 1, 2) / 2 + 0.5).clamp(0, 1).contiguous()


ALL = ("wave_conv1d", "groupnorm", "groupnorm_stats", "groupnorm_fold_linear", "layernorm", "row_stats", "layernorm_fp8", "quantize_fp8", "gemm", "geglu", "ff_fused", "tblock_fused", "alloc_vt", "gemm_split", "key_norm_max", "attention",
       "temporal_attention", "small_kv_attention", "audio_xattn_pack", "audio_xattn", "upsample_conv_phases", "add_residual_f32", "add_row_bias", "gather_latents", "cfg_combine", "pack_rows", "combine_units", "overlap_ddim_step",
       "ncfhw_to_nhwc", "nhwc_to_ncfhw", "vae_postprocess")


def install(monkeypatch, ops):
    for name in ALL:
        monkeypatch.setattr(ops, name, globals()[name])
