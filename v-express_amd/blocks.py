"""Block-level forward functions over channels-last tokens, composed purely of libvexpress_hip ops.

Every function cites the reference code it reproduces (paths relative to tencent-ailab/V-Express).
Activations: bf16 `[frames, H*W, C]` (frames = b*f, batch-major like the reference's `(b f)` flattening).
No einops / permutes ever run: the reference's `b c f h w <-> (b f) c h w <-> (b f) hw c <-> (b hw) f c`
ping-pong (resnet.py:13-15,24-26; transformer_3d.py:115,165; motion_module.py:151,180,361-363,386)
collapses to this one resident layout.
"""
import torch

from . import ops
from .ops import ConvGeom


def resnet_block(P, x, frames, H, W, *, groups, eps, temb=None, rows_per_group=0, skip=None, items=None):
    with ops.frame_rows(H * W, items=items):
        return _resnet_block(P, x, frames, H, W, groups=groups, eps=eps, temb=temb, rows_per_group=rows_per_group,
                             skip=skip)


def _resnet_block(P, x, frames, H, W, *, groups, eps, temb=None, rows_per_group=0, skip=None):
    """ResnetBlock3D.forward (modules/resnet.py:217-251) / diffusers ResnetBlock2D.
    x: [frames, HW, C1]; skip: optional [frames, HW, C2] consumed as the channel concat
    torch.cat([x, skip], dim=1) (modules/unet_3d_blocks.py:694,831) without materialising it;
    temb: fp32 [b, Cout] view = time_emb_proj(silu(emb)) rows (resnet.py:225-233)."""
    hw = H * W
    c1 = x.shape[-1]
    cout = P.conv1.w.shape[0]
    g3 = ConvGeom(frames, H + 2, W + 2, 3, 3, 1, 0)
    hp = (H + 2) * (W + 2)
    # GroupNorm writes into a zero-bordered (H+2)x(W+2) image, so the 3x3 convs are pad-0 convs (vx_gemm fast path).
    # (The one-pass form - GroupNorm + SiLU applied in the convolution's A path - was built and measured in round 5:
    # correct, 0.8 % slower; it lives in tools/conv3/ with its emulator, not in the shipped ABI: LABNOTES 12.3.)
    n = ops.groupnorm(x, P.norm1.g, P.norm1.b, frames=frames, hw=hw, groups=groups, eps=eps, silu=True, x2=skip,
                      pad_hw=(H, W))
    # conv1's epilogue leaves norm2's partial sums on h (ops.gemm(gn=...)): norm2 is an apply pass only
    h = ops.gemm(n.view(frames * hp, -1), P.conv1.w, P.conv1.b, geom=g3, rowbias=temb, rows_per_group=rows_per_group,
                 gn=(groups, hw))
    if P.shortcut is not None:
        sc = ops.gemm(x.view(frames * hw, c1), P.shortcut.w, P.shortcut.b,
                      a2=None if skip is None else skip.view(frames * hw, -1))
    else:
        if skip is not None:
            raise ValueError("concat input needs a conv_shortcut")
        sc = x.view(frames * hw, c1)
    # every resnet output is read by a GroupNorm next (Transformer3DModel.norm or the motion module's norm)
    n2 = ops.groupnorm(ops.keep_gn(h.view(frames, hw, cout), h), P.norm2.g, P.norm2.b, frames=frames, hw=hw,
                       groups=groups, eps=eps, silu=True, pad_hw=(H, W))
    out = ops.gemm(n2.view(frames * hp, cout), P.conv2.w, P.conv2.b, geom=g3, residual=sc, gn=(groups, hw))
    return ops.keep_gn(out.view(frames, hw, cout), out)


def downsample(P, x, frames, H, W, gn_groups=0):
    """Downsample3D: conv3x3 stride 2 pad 1 (modules/resnet.py:106-118).  gn_groups: the next block's first resnet
    normalises this output (norm1): leave its partial sums on the tensor."""
    g = ConvGeom(frames, H, W, 3, 3, 2, 1)
    out = ops.gemm(x.view(frames * H * W, -1), P.w, P.b, geom=g,
                   gn=(gn_groups, g.h_out * g.w_out) if gn_groups else None)
    return ops.keep_gn(out.view(frames, g.h_out * g.w_out, -1), out), g.h_out, g.w_out


def upsample(P, x, frames, H, W, items=None):
    """Upsample3D: nearest x2 + conv3x3 (modules/resnet.py:53-90): as four 2x2 convolutions over the original image where
    `ops.upsample_phases_applies` (2.25x fewer FLOPs), else one 3x3 convolution with the upsampling fused into its gather.
    items: independent batch items in `frames` (CFG halves; frames for per-frame models) - the kernel choice of the four
    launches is a function of ONE item's rows (ops.frame_rows), never of the batch."""
    if P.get("phases") is not None and ops.upsample_phases_applies(H, W, x.shape[-1]):
        with ops.frame_rows(H * W, items=items):
            out = ops.upsample_conv_phases(x.reshape(frames, H * W, -1), P.phases.w, P.phases.b, frames=frames, H=H, W=W)
        return out, 2 * H, 2 * W
    g = ConvGeom(frames, H, W, 3, 3, 1, 1, upsample=1)
    out = ops.gemm(x.view(frames * H * W, -1), P.w, P.b, geom=g)
    return out.view(frames, g.h_out * g.w_out, -1), g.h_out, g.w_out


def _fold_on(F):
    """A folded LayerNorm is used when the weights carry one, the switch is on and the projections are not in fp8 mode
    (whose quantiser is fused with its own LayerNorm)."""
    return F is not None and ops.LN_FOLD[0] and not ops.FP8_PROJ[0]


def _self_attention(A, ln, h, *, seqs, n_tok, heads, fold=None, item_bias=None, rows_per_item=0, stats_out=None):
    """diffusers Attention as self-attention: fused QKV GEMM whose epilogue also emits V^T, flash attention,
    out-projection with the residual add fused (in place on h).
    fold = (folded weights, row statistics): `ln` is then the UN-normalised h and the LayerNorm runs inside the GEMM.
    item_bias: float32 [items, C] added to the out-projection per batch item (rows_per_item rows each) in the same
    epilogue - the constant contributions of the blocks that follow for that item (see _spatial_transformer_read)."""
    m, c = ln.shape
    d = c // heads
    vt = ops.alloc_vt(seqs, heads, d, n_tok, ln.device)
    qk_ring = fold is not None and ops.qk_on_ring(m, c)
    if not qk_ring:
        q = torch.empty((m, c), device=ln.device, dtype=ops.BF16)
        k = torch.empty((m, c), device=ln.device, dtype=ops.BF16)
    if qk_ring:
        # Q | K as ONE [m, 2C] GEMM on the persistent ring kernel (q, k = column views of its output; the attention
        # kernels take row strides), V^T alone through the SPLIT epilogue of the classic tiles: the fused 3C-wide SPLIT
        # launch runs on the classic 256 x 320 tile at ~500 TFLOP/s at the 64x64 level, the ring kernel at ~700
        F, stats = fold
        qk = ops.gemm(ln, F.w[:2 * c], F.b[:2 * c], ln=(stats, F.s[:2 * c]))
        q, k = qk[:, :c], qk[:, c:]
        ops.gemm_split(ln, F.w[2 * c:], F.b[2 * c:], [("vt", vt)], part_cols=c, seq_len=n_tok, head_dim=d,
                       ln=(stats, F.s[2 * c:]))
    elif fold is not None:
        F, stats = fold
        ops.gemm_split(ln, F.w, F.b, [("rows", q), ("rows", k), ("vt", vt)], part_cols=c, seq_len=n_tok, head_dim=d,
                       ln=(stats, F.s))
    else:
        ops.gemm_split(ln, ops.proj_weight(ln, A.wqkv), A.bqkv, [("rows", q), ("rows", k), ("vt", vt)], part_cols=c,
                       seq_len=n_tok, head_dim=d)
    a = ops.attention(q, k, vt, batch=seqs, heads=heads, n_q=n_tok, n_kv=n_tok, head_dim=d,
                      k_prescaled=bool(A.get("k_prescaled")))
    if isinstance(ln, ops.Fp8Rows):
        a = ops.quantize_fp8(a)
    ops.gemm(a, ops.proj_weight(a, A.out.w), A.out.b, residual=h, out=h, rowbias=item_bias,
             rows_per_group=rows_per_item, stats_out=stats_out)


def _ff_fold_on(P):
    return P.get("ln_ff") is not None and ops.LN_FOLD[0]


def _norm_proj_in(P, x, frames, hw, groups, stats_out=None):
    """h = proj_in(GroupNorm(x)) (modules/transformer_3d.py:124-126, modules/motion_module.py:156-158).  Where
    `ops.gn_fold_applies` (the 64x64 level) the normalised tensor is never materialised: the statistics pass alone,
    one scaled weight copy + bias row per frame (`ops.groupnorm_fold_linear`), and the GEMM reads the raw x with
    per-frame weights.  stats_out: see ops.gemm (row statistics of h for the LayerNorm the next GEMM folds)."""
    c = x.shape[-1]
    m = frames * hw
    G = P.get("gn_fold")
    if G is not None and ops.gn_fold_applies(m, hw, c, G.w.shape[0]):
        ws, slices = ops.groupnorm_stats(x, frames=frames, hw=hw, groups=groups)      # the producer's own when it left them
        w_f, b_f = ops.groupnorm_fold_linear(ws, G.g, G.w, G.bb, frames=frames, hw=hw, groups=groups, eps=1e-6,
                                             slices=slices)
        return ops.gemm(x.view(m, c), w_f, None, rowbias=b_f, rows_per_group=hw, w_group_rows=hw, stats_out=stats_out)
    n = ops.groupnorm(x, P.norm.g, P.norm.b, frames=frames, hw=hw, groups=groups, eps=1e-6, silu=False)
    return ops.gemm(n.view(m, c), P.proj_in.w, P.proj_in.b, stats_out=stats_out)


def _feed_forward(P, h, stats=None, proj=None):
    """h += FF(LN(h)): GEGLU fused in the first GEMM's epilogue (with the LayerNorm folded into that GEMM when the
    weights carry the fold), residual in the second's.  stats: the row statistics of h when its producer emitted them.
    proj = (residual, gn): the block's proj_out follows (out = proj_out(h) + residual, nothing else reads h; residual None:
    float32 rows without the residual, see _motion_module): where
    `ops.ff_proj_fold_applies` the second GEMM and proj_out run as ONE dual-source launch over [h | g] with the load-time
    fold weights.fold_ff_proj, and the function RETURNS out (None otherwise: the caller runs proj_out itself).
    ops.FF_SLAB_BYTES: the [rows, 4C] GEGLU result of a whole launch (335 MB at the 64x64 level) is larger than the 256 MB
    Infinity Cache, so the second GEMM re-reads all of it from HBM; run in row slabs whose intermediate fits, each slab's
    second GEMM right behind its first, it reads what the first just wrote.  Rows are independent: same bits."""
    F = P.get("ln_ff")
    fold = F is not None and ops.LN_FOLD[0]
    m, c = h.shape
    n_hidden = (F.w if fold else P.ff.w1).shape[0] // 2
    slabs = 1
    if ops.FF_SLAB_BYTES[0] > 0:
        # whole 256-row tiles per slab and, when the launches run on the persistent kernel, at least one tile per CU in
        # the narrower (second) GEMM: slab rows = a multiple of 256 * 256 / (c / 320)
        unit = 256 * max(1, 256 // max(1, c // 320))
        want = -(-(m * n_hidden * 2) // ops.FF_SLAB_BYTES[0])
        while slabs < want and m % (2 * slabs * unit) == 0:
            slabs *= 2
    if fold and stats is None:
        stats = ops.row_stats(h)
    if fold and ops.ff_fused_applies(m, c, n_hidden):
        ops.ff_fused(h, F.w, F.b, F.s, stats, P.ff.out.w, P.ff.out.b)      # one launch, no [m, 4C] intermediate
        return None
    if proj is not None and fold and slabs == 1 and P.get("ff_proj") is not None and ops.ff_proj_fold_applies(m, c, n_hidden):
        residual, gn = proj
        g = ops.geglu(h, F.w, F.b, ln=(stats, F.s))
        # residual None: the caller adds it later (frame shards) - float32 rows = accumulator + bias, unrounded
        return ops.gemm(h, P.ff_proj.w, P.ff_proj.b, a2=g, residual=residual, gn=gn, out_f32=residual is None)
    rows = m // slabs
    for i in range(slabs):
        hs = h[i * rows:(i + 1) * rows]
        if fold:
            g = ops.geglu(hs, F.w, F.b, ln=(stats[i * rows:(i + 1) * rows], F.s))
        else:
            ln = ops.layernorm(hs, P.norm3.g if "norm3" in P else P.ff_norm.g, P.norm3.b if "norm3" in P else P.ff_norm.b)
            g = ops.geglu(ln, P.ff.w1, P.ff.b1)
        ops.gemm(g, P.ff.out.w, P.ff.out.b, residual=hs, out=hs)
    return None


def audio_kv(P, ehs):
    """K | V of the audio cross-attention (attn2) of one transformer block: ehs [b*f*n_ctx, 768] -> [b*f*n_ctx, 2C].
    Depends on the audio tokens and the weights only, so the denoising loop computes it once per clip and window
    (UNet3DConditionModel.precompute_audio_kv) instead of once per block per DDIM step."""
    return ops.gemm(ehs, P.attn2.wkv)


def _audio_fold(P, kv, f0, f1, n_ctx, heads):
    """The vx_audio_xattn operands of frames f0 .. f1-1 of `kv`, built at first use and kept on the kv tensor object (which
    the denoising loop holds for all DDIM steps of a window: unet_3d.precompute_audio_kv)."""
    cache = getattr(kv, "_vx_audio_fold", None)
    if cache is None:
        cache = {}
        try:
            kv._vx_audio_fold = cache
        except AttributeError:
            pass
    fold = cache.get((f0, f1))
    if fold is None:
        fold = cache[(f0, f1)] = ops.audio_xattn_pack(kv[f0 * n_ctx:f1 * n_ctx], P.ln_q2.w, P.ln_q2.b, P.attn2.out.w,
                                                      frames=f1 - f0, n_ctx=n_ctx, heads=heads)
    return fold


def spatial_transformer_read(P, x, *, b, f, H, W, heads, groups, ehs, bank, w_ref, w_aud, kv=None, audio_zero=None):
    with ops.frame_rows(H * W, items=b):
        return _spatial_transformer_read(P, x, b=b, f=f, H=H, W=W, heads=heads, groups=groups, ehs=ehs, bank=bank,
                                         w_ref=w_ref, w_aud=w_aud, kv=kv, audio_zero=audio_zero)


def _spatial_transformer_read(P, x, *, b, f, H, W, heads, groups, ehs, bank, w_ref, w_aud, kv=None, audio_zero=None):
    """Transformer3DModel.forward (modules/transformer_3d.py:103-169) with the block forward patched by
    ReferenceAttentionControl in *read* mode (modules/mutual_self_attention.py:176-267).
    x: [b*f, HW, C]; ehs: bf16 [b*f*n_ctx, 768] audio tokens; bank: list over the b batch rows of
    None (all-zero bank -> the attention output is exactly to_out.bias) or (k, vt, kmax)."""
    frames, hw, c = b * f, H * W, x.shape[-1]
    m = frames * hw
    x2d = x.view(m, c)
    # Row statistics of the residual stream h for the LayerNorms folded into the q / qkv / GEGLU projections: every
    # GEMM that writes rows of h also writes their (mean, rstd) into `st` when the next reader of those rows folds its
    # LayerNorm (at the 64x64 level straight from the epilogue's registers: ops.gemm(stats_out=...))
    f_qkv, f_q15, f_q2 = _fold_on(P.get("ln_qkv")), _fold_on(P.get("ln_q15")), _fold_on(P.get("ln_q2"))
    f_ff = _ff_fold_on(P)
    st = ops.stats_buffer(m, c, x.device) if (f_qkv or f_q15 or f_q2 or f_ff) else None
    h = _norm_proj_in(P, x, frames, hw, groups, stats_out=st if f_qkv else None)
    # A batch item without a bank (the unconditional CFG half) gets exactly w_ref * attn1_5.to_out.bias from block 1.5,
    # and - when its audio tokens are all zero as well - exactly w_aud * attn2.to_out.bias from block 2 (SURVEY.md App.
    # E4); nothing reads that item's h in between, so both constants ride in the epilogue of the attn1 out-projection
    # as a per-item row bias instead of two read-modify-write passes over h (vx_add_row_bias) per block.
    d = c // heads
    rows = f * hw
    fold_ref = [bank[bi] is None for bi in range(b)]
    fold_aud = [ops.FOLD_ZERO_AUDIO[0] and fold_ref[bi] and audio_zero is not None and bool(audio_zero[bi])
                for bi in range(b)]
    item_bias = None
    if any(fold_ref):
        key = ("item_bias", tuple(fold_ref), tuple(fold_aud), float(w_ref), float(w_aud))
        item_bias = P.get(key)
        if item_bias is None:                      # built once per (CFG pattern, weights): a few tiny launches
            item_bias = torch.zeros((b, c), device=x.device, dtype=torch.float32)
            for bi in range(b):
                if fold_ref[bi]:
                    item_bias[bi] = P.attn1_5.out.b * w_ref + (P.attn2.out.b * w_aud if fold_aud[bi] else 0.0)
            P[key] = item_bias
    # 1. self-attention (:177-184)
    st1 = st if (f_q15 or f_q2 or f_ff) else None
    if f_qkv:
        _self_attention(P.attn1, h, h, seqs=frames, n_tok=hw, heads=heads, fold=(P.ln_qkv, st),
                        item_bias=item_bias, rows_per_item=rows, stats_out=st1)
    else:
        ln = ops.proj_layernorm(h, P.norm1.g, P.norm1.b)
        _self_attention(P.attn1, ln, h, seqs=frames, n_tok=hw, heads=heads, item_bias=item_bias, rows_per_item=rows,
                        stats_out=st1)
    # 1.5 reference attention (:186-224): K/V = bank of the batch row, shared by its f frames
    for bi in range(b):
        hb = h[bi * rows:(bi + 1) * rows]
        sb = None if st is None else st[bi * rows:(bi + 1) * rows]
        if bank[bi] is None:
            pass                                   # folded into the attn1 out-projection above
        else:
            kref, vtref, kmax = bank[bi]
            with ops.frame_rows(hw, items=1):          # these launches cover ONE batch item
                if f_q15:
                    q = ops.gemm(hb, P.ln_q15.w, P.ln_q15.b, ln=(sb, P.ln_q15.s))
                else:
                    ln = ops.proj_layernorm(hb, P.norm1_5.g, P.norm1_5.b)
                    q = ops.gemm(ln, ops.proj_weight(ln, P.attn1_5.wq))
                a = ops.proj_input(ops.attention(q, kref, vtref, batch=f, heads=heads, n_q=hw, n_kv=kref.shape[0],
                                                 head_dim=d, q_per_kv=f, kmax=kmax,
                                                 k_prescaled=bool(P.attn1_5.get("k_prescaled"))))
                ops.gemm(a, ops.proj_weight(a, P.attn1_5.out.w), P.attn1_5.out.b, residual=hb, alpha=w_ref, out=hb,
                         stats_out=sb if (f_q2 or f_ff) else None)
    # 2. audio cross-attention (:227-244).  A batch row whose audio tokens are ALL ZERO (the unconditional CFG half:
    # torch.zeros_like, pipelines/v_express_pipeline.py:403-405) has K = V = 0 (to_k / to_v carry no bias): every
    # score is 0, the softmax is uniform, the weighted sum of V is exactly 0 and the block adds exactly
    # audio_attention_weight * to_out.bias - the same argument as the all-zero bank of 1.5.
    n_ctx = ehs.shape[0] // frames
    if kv is None:
        kv = audio_kv(P, ehs)
    # One launch per (block, batch item with audio): q-projection, 5-key attention and out-projection collapse into two
    # 48-column products with per-frame operands (ops.audio_xattn; built once per kv tensor, i.e. once per clip and window)
    ax = f_q2 and ops.audio_xattn_applies(c, heads, n_ctx, hw)
    if ax and (audio_zero is None or not any(audio_zero)):
        ops.audio_xattn(h, st, _audio_fold(P, kv, 0, frames, n_ctx, heads), P.attn2.out.b, w_aud, rows_per_frame=hw,
                        stats_out=st if f_ff else None)
    elif audio_zero is None or not any(audio_zero):
        if f_q2:
            q = ops.gemm(h, P.ln_q2.w, P.ln_q2.b, ln=(st, P.ln_q2.s))
        else:
            ln = ops.proj_layernorm(h, P.norm2.g, P.norm2.b)
            q = ops.gemm(ln, ops.proj_weight(ln, P.attn2.wq))
        a = ops.proj_input(ops.small_kv_attention(q, kv, batch=frames, n_q=hw, n_kv=n_ctx, heads=heads, head_dim=d))
        ops.gemm(a, ops.proj_weight(a, P.attn2.out.w), P.attn2.out.b, residual=h, alpha=w_aud, out=h,
                 stats_out=st if f_ff else None)
    else:
        kvr = f * n_ctx
        for bi in range(b):
            hb = h[bi * rows:(bi + 1) * rows]
            sb = None if st is None else st[bi * rows:(bi + 1) * rows]
            if audio_zero[bi]:
                if not fold_aud[bi]:
                    ops.add_row_bias(hb, P.attn2.out.b, w_aud)
                    if f_ff:
                        ops.row_stats(hb, out=sb)      # the rows changed after their producer's statistics
                continue
            if ax:
                ops.audio_xattn(hb, sb, _audio_fold(P, kv, bi * f, (bi + 1) * f, n_ctx, heads), P.attn2.out.b, w_aud,
                                rows_per_frame=hw, stats_out=sb if f_ff else None)
                continue
            with ops.frame_rows(hw, items=1):
                if f_q2:
                    q = ops.gemm(hb, P.ln_q2.w, P.ln_q2.b, ln=(sb, P.ln_q2.s))
                else:
                    ln = ops.proj_layernorm(hb, P.norm2.g, P.norm2.b)
                    q = ops.gemm(ln, ops.proj_weight(ln, P.attn2.wq))
                a = ops.proj_input(ops.small_kv_attention(q, kv[bi * kvr:(bi + 1) * kvr], batch=f, n_q=hw, n_kv=n_ctx,
                                                          heads=heads, head_dim=d))
                ops.gemm(a, ops.proj_weight(a, P.attn2.out.w), P.attn2.out.b, residual=hb, alpha=w_aud, out=hb,
                         stats_out=sb if f_ff else None)
    # 3. feed-forward (:247)
    out = _feed_forward(P, h, st if f_ff else None, proj=(x2d, (groups, hw)))
    if out is None:
        out = ops.gemm(h, P.proj_out.w, P.proj_out.b, residual=x2d, gn=(groups, hw))  # read next by the motion module's norm
    return ops.keep_gn(out.view(frames, hw, c), out)


def spatial_transformer_write(P, x, *, frames, H, W, heads, groups, ehs):
    with ops.frame_rows(H * W):
        return _spatial_transformer_write(P, x, frames=frames, H=H, W=W, heads=heads, groups=groups, ehs=ehs)


def _spatial_transformer_write(P, x, *, frames, H, W, heads, groups, ehs):
    """Transformer2DModel.forward (modules/transformer_2d.py:216-399) with the BasicTransformerBlock forward
    patched in *write* mode (modules/mutual_self_attention.py:145-174, FF tail :269-284).
    Returns (output, bank) with bank = norm2(h + attn1(norm1 h)) as [frames*HW, C]."""
    hw, c = H * W, x.shape[-1]
    m = frames * hw
    d = c // heads
    n = ops.groupnorm(x, P.norm.g, P.norm.b, frames=frames, hw=hw, groups=groups, eps=1e-6, silu=False)
    h = ops.gemm(n.view(m, c), P.proj_in.w, P.proj_in.b)
    ln = ops.layernorm(h, P.norm1.g, P.norm1.b)
    _self_attention(P.attn1, ln, h, seqs=frames, n_tok=hw, heads=heads)
    bank = ops.layernorm(h, P.norm2.g, P.norm2.b)
    n_ctx = ehs.shape[0] // frames
    q = ops.gemm(bank, P.attn2.wq)
    kv = ops.gemm(ehs, P.attn2.wkv)
    a = ops.small_kv_attention(q, kv, batch=frames, n_q=hw, n_kv=n_ctx, heads=heads, head_dim=d)
    ops.gemm(a, P.attn2.out.w, P.attn2.out.b, residual=h, out=h)
    _feed_forward(P, h)
    out = ops.gemm(h, P.proj_out.w, P.proj_out.b, residual=x.view(m, c))
    return out.view(frames, hw, c), bank


def motion_module(P, x, *, b, f, H, W, heads, groups, shard=None, gn_next=False):
    with ops.frame_rows(H * W, items=b):
        return _motion_module(P, x, b=b, f=f, H=H, W=W, heads=heads, groups=groups, shard=shard, gn_next=gn_next)


def _motion_module(P, x, *, b, f, H, W, heads, groups, shard=None, gn_next=False):
    """VanillaTemporalModule -> TemporalTransformer3DModel.forward (modules/motion_module.py:146-182), one
    TemporalTransformerBlock (:236-259): 2x [LN, +pe, QKV, attention over f, out-proj + residual], LN, GEGLU FF.
    The additive sinusoid table goes through the LayerNorm kernel (pe enters Q, K and V: :365-366).
    shard (distributed.FrameShard): x holds only this rank's f frames of the window.  The GroupNorm (per-frame
    statistics) runs on them; the temporal transformer - per token except the attention over the frame axis - runs on
    all shard.size * f frames of this rank's pixel slice between two all-to-alls; the output projection and the
    residual add are per token again and run back in the frame-shard layout.
    gn_next: the output is read by a GroupNorm over these channels next (the following resnet's norm1 without a skip
    concat, or conv_norm_out): the out-projection leaves that GroupNorm's partial sums on the tensor."""
    frames, hw, c = b * f, H * W, x.shape[-1]
    d = c // heads
    f_all, hw_t = f, hw
    if shard is not None:
        f_all, hw_t = f * shard.size, hw // shard.size
    m = b * f_all * hw_t
    # row statistics of h for the folded LayerNorms: written by the GEMM that produces the rows (see the spatial block)
    folds = [_fold_on(A.get("ln_qkv")) for A in P.attn] + [_ff_fold_on(P)]
    # attention blocks that run as ONE launch (ops.tblock_fused: the 64x64 level): statistics in and out like the GEMMs
    fused = [folds[i] and ops.tblock_fused_applies(c, heads, f_all, hw_t) for i in range(len(P.attn))] + [False]
    wants = folds                                                            # consumers of row statistics
    st = ops.stats_buffer(m, c, x.device) if any(wants) else None
    if shard is not None:
        n = ops.groupnorm(x, P.norm.g, P.norm.b, frames=frames, hw=hw, groups=groups, eps=1e-6, silu=False)
        n = shard.to_pixel_shard(n, b, f)
        h = ops.gemm(n.view(m, c), P.proj_in.w, P.proj_in.b, stats_out=st if wants[0] else None)
    else:
        h = _norm_proj_in(P, x, frames, hw, groups, stats_out=st if wants[0] else None)
    have_st = wants[0]                  # st holds the statistics of the current h
    for i, A in enumerate(P.attn):
        if fused[i]:
            ops.tblock_fused(h, A.ln_qkv.w, A.ln_qkv.b, A.ln_qkv.s, A.pe_rows, A.attn.out.w, A.attn.out.b, b=b, f=f_all,
                             hw=hw_t, heads=heads, stats=st if have_st else None, stats_out=st if wants[i + 1] else None)
            have_st = wants[i + 1]
            continue
        if folds[i]:
            if not have_st:
                st = ops.row_stats(h)
            key = ("pe_rows_tiled", b, f_all)
            if key not in A:                   # [b * f_all, 3C] float32: frame (m // hw_t) % f_all of the table
                A[key] = A.pe_rows[:f_all].repeat(b, 1).contiguous()
            qkv = ops.gemm(h, A.ln_qkv.w, A.ln_qkv.b, rowbias=A[key], rows_per_group=hw_t, ln=(st, A.ln_qkv.s))
        else:
            ln = ops.proj_layernorm(h, A.norm.g, A.norm.b, add=A.pe, add_rows_per_entry=hw_t, add_entries=f_all)
            qkv = ops.gemm(ln, ops.proj_weight(ln, A.attn.wqkv), A.attn.bqkv)
        a = ops.proj_input(ops.temporal_attention(qkv, b=b, f=f_all, hw=hw_t, heads=heads, head_dim=d))
        ops.gemm(a, ops.proj_weight(a, A.attn.out.w), A.attn.out.b, residual=h, out=h,
                 stats_out=st if wants[i + 1] else None)
        have_st = wants[i + 1]
    # proj_out folded into the feed-forward's second linear (weights.fold_ff_proj).  Frame shards: the folded GEMM runs in the
    # pixel-shard layout into FLOAT32 (accumulator + bias, exactly what the unsharded epilogue holds before its residual
    # add), those rows go through the all-to-all, and the residual add + the one rounding happen in the frame-shard layout
    # (ops.add_residual_f32): the same bits as the unsharded launch, for twice the bytes of one all-to-all.
    out = _feed_forward(P, h, st if folds[-1] and have_st else None,
                        proj=((None if shard is not None else x.view(frames * hw, c)),
                              ((groups, hw) if gn_next and shard is None else None)) if ops.FF_PROJ_FOLD_MM[0] else None)
    if out is not None and shard is not None:
        y32 = shard.to_frame_shard(out.view(b * f_all, hw_t, c), b, f).view(frames * hw, c)
        out = ops.add_residual_f32(x.view(frames * hw, c), y32)
    elif out is None:
        if shard is not None:
            h = shard.to_frame_shard(h.view(b * f_all, hw_t, c), b, f).view(frames * hw, c)
        out = ops.gemm(h, P.proj_out.w, P.proj_out.b, residual=x.view(frames * hw, c), gn=(groups, hw) if gn_next else None)
    return ops.keep_gn(out.view(frames, hw, c), out)


def bank_kv(A, bank_tokens, heads):
    """Per-clip precompute of the reference-attention K / V^T from a writer bank (step-invariant):
    k = bank Wk^T, v = bank Wv^T of the reader's attn1_5 (modules/mutual_self_attention.py:204-224)."""
    n, c = bank_tokens.shape
    d = c // heads
    k = torch.empty((n, c), device=bank_tokens.device, dtype=ops.BF16)
    vt = ops.alloc_vt(1, heads, d, n, bank_tokens.device)
    ops.gemm_split(bank_tokens, A.wkv, None, [("rows", k), ("vt", vt)], part_cols=c, seq_len=n, head_dim=d)
    # max key norm per head: the bound table of the bounded-softmax attention kernel (step-invariant like K itself)
    return k, vt, ops.key_norm_max(k, kv_batches=1, heads=heads, n_kv=n, head_dim=d)
