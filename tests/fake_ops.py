"""Torch/CPU stand-ins for a few `v_express_amd.ops` entry points — TEST INFRASTRUCTURE for the CPU suite only.

They let the *host composition* of a model (weight re-layouts, strided window views, buffer plumbing, call order) be
checked against the oracle in the dev container, where no GPU exists.  Same signatures and output dtypes (bf16 rounding
at every kernel boundary) as the real wrappers; the arithmetic inside is plain fp32 torch.  Never imported by the
package; the GPU tests run the same model code on the real kernels.
"""
import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


def _act(y, act):
    return F.silu(y) if act == 1 else F.gelu(y) if act == 2 else y


def wave_conv1d(wave, wt, stride):
    taps, _ = wt.shape
    return (wave.unfold(0, taps, stride) @ wt).to(BF16)


def groupnorm(x1, gamma, beta, *, frames, hw, groups, eps, silu, x2=None, out=None, pad_hw=None):
    assert x2 is None and out is None and pad_hw is None and x1.is_contiguous()
    x = x1.float().view(frames, hw, -1).transpose(1, 2)
    y = F.group_norm(x, groups, gamma, beta, eps)
    return _act(y, int(silu)).transpose(1, 2).contiguous().to(BF16)


def layernorm(x, gamma, beta, eps=1e-5, *, add=None, add_rows_per_entry=1, add_entries=1, out=None):
    assert add is None and x.dtype == BF16
    y = F.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps).to(BF16)
    if out is not None:
        out.copy_(y)
        return out
    return y


def gemm(a, w, bias=None, *, geom=None, a2=None, residual=None, alpha=1.0, act=0, rowbias=None, rows_per_group=0,
         out=None, out_f32=False):
    assert geom is None and a2 is None and rowbias is None and a.dtype == BF16 and w.dtype == BF16
    assert a.stride(-1) == 1 and a.stride(0) % 8 == 0 and w.is_contiguous() and a.shape[1] == w.shape[1]
    y = a.float() @ w.float().t()
    if bias is not None:
        assert bias.dtype == torch.float32
        y = y + bias
    y = _act(y, act)
    if residual is not None:
        y = residual.float() + alpha * y
    elif alpha != 1.0:
        y = alpha * y
    y = y if out_f32 else y.to(BF16)
    if out is not None:
        assert out.shape == y.shape and out.stride(-1) == 1
        out.copy_(y)
        return out
    return y


def alloc_vt(seqs, heads, head_dim, n, device):
    return torch.zeros((seqs, heads, head_dim, (n + 7) // 8 * 8), device=device, dtype=BF16)


def gemm_split(a, w, bias, parts, *, part_cols, seq_len=0, head_dim=0, geom=None):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    for i, (kind, t) in enumerate(parts):
        col = y[:, i * part_cols:(i + 1) * part_cols]
        if kind == "rows":
            t.copy_(col.to(BF16))
        else:
            seqs, heads = t.shape[0], t.shape[1]
            t[..., :seq_len] = col.view(seqs, seq_len, heads, head_dim).permute(0, 2, 3, 1).to(BF16)


def attention(q, k, vt, *, batch, heads, n_q, n_kv, head_dim, q_per_kv=1, out=None, kmax=None):
    assert q_per_kv == 1
    qh = q.float().view(batch, n_q, heads, head_dim).transpose(1, 2)
    kh = k.float().view(batch, n_kv, heads, head_dim).transpose(1, 2)
    vh = vt[..., :n_kv].float().transpose(-1, -2)
    o = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(batch * n_q, heads * head_dim).to(BF16)
    if out is not None:
        out.copy_(o)
        return out
    return o


def install(monkeypatch, ops):
    for name in ("wave_conv1d", "groupnorm", "layernorm", "gemm", "alloc_vt", "gemm_split", "attention"):
        monkeypatch.setattr(ops, name, globals()[name])
