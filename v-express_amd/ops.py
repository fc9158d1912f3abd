"""Tensor-level wrappers around the C ABI (include/vexpress_hip.h).

PyTorch is used for device memory and streams only: every wrapper passes raw device pointers, explicit
shapes/strides and `torch.cuda.current_stream()` to libvexpress_hip.so.  Activations are bf16
channels-last tokens `[frames, H*W, C]`; weights are pre-laid-out by `weights.py`.
"""
import ctypes as C
import logging
import os

import torch

from . import lib as L

class _CurrentLib:
    """`_lib.vx_*` resolves to the library of the element type in force (lib.ELEM: bfloat16 or IEEE half; lib.element_type)."""

    def __getattr__(self, name):
        return getattr(L.current(), name)


_lib = _CurrentLib()


def __getattr__(name):
    # `ops.BF16` = the element dtype in force (the name predates the IEEE-half library: torch.bfloat16 or torch.float16)
    if name == "BF16":
        return L.ELEM[0]
    raise AttributeError(name)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Weight of the launches that follow in a whole clip (bench.py's instrumented leg runs 1 of the clip's DDIM steps and
# decodes a few of its frames: it sets the factor before each part); stored with every profile record
PROFILE_WEIGHT = [1.0]


class GemmProfile:
    """Optional per-launch HIP-event timing of vx_gemm (bench.py's roofline leg).  Events are recorded on the
    same stream the kernels are launched on (torch's current stream)."""
    active = None

    def __init__(self):
        self.records = []          # (start_event, end_event, flops, tile_key, shape, kernel symbol, clip weight)
        self.bytes_of = {}         # id(start_event) -> algorithmic bytes of that launch (operands read once + output)
        self.shape_bytes = {}      # (m, n, k, tile key) -> algorithmic bytes summed over its launches

    def __enter__(self):
        GemmProfile.active = self
        return self

    def __exit__(self, *a):
        GemmProfile.active = None

    def summary(self):
        torch.cuda.synchronize()
        by = {}
        for s, e, fl, key, *_ in self.records:
            d = by.setdefault(key, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += s.elapsed_time(e) * 1e-3
            d[2] += fl
            d[3] += self.bytes_of.get(id(s), 0.0)
        return {k: dict(launches=v[0], seconds=v[1], flops=v[2], bytes=v[3]) for k, v in by.items()}

    def by_symbol(self):
        """{kernel instantiation as rocprofv3 names it: dict(launches, seconds, flops, bytes)} - pairs one to one with
        the rows of a committed kernel trace / counter file of the same library build."""
        torch.cuda.synchronize()
        by = {}
        for rec in self.records:
            s, e, fl, sym = rec[0], rec[1], rec[2], rec[5]
            d = by.setdefault(sym, dict(launches=0, seconds=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["seconds"] += s.elapsed_time(e) * 1e-3
            d["flops"] += fl
            d["bytes"] += self.bytes_of.get(id(s), 0.0)
        return by

    def by_shape(self):
        """{(m, n, k, kernel): [launches, seconds, flops]} sorted by time (tools / tuning)."""
        torch.cuda.synchronize()
        by = {}
        for s, e, fl, key, shape, *_ in self.records:
            d = by.setdefault(shape + (key,), [0, 0.0, 0.0])
            d[0] += 1
            d[1] += s.elapsed_time(e) * 1e-3
            d[2] += fl
        return dict(sorted(by.items(), key=lambda kv: -kv[1][1]))


class OpProfile:
    """Optional per-call HIP-event timing of every non-GEMM wrapper with its algorithmic bytes and - for the MFMA kernels
    (attention, the fused feed-forward / temporal blocks) - its algorithmic FLOPs and the kernel instantiation it launched
    (vx_last_kernel): bench.py's `roofline.hbm_kernels` table (achieved GB/s against the 8 TB/s of HBM3E) and,
    together with GemmProfile, the choice of the dominant kernel over ALL kernels.  Same stream as the launches."""
    active = None

    def __init__(self):
        self.records = []          # (name, algorithmic bytes, start_event, end_event, flops, kernel symbol or name, clip weight)

    def __enter__(self):
        OpProfile.active = self
        return self

    def __exit__(self, *a):
        OpProfile.active = None

    def summary(self):
        torch.cuda.synchronize()
        by = {}
        for name, nbytes, s, e, fl, *_ in self.records:
            d = by.setdefault(name, dict(launches=0, seconds=0.0, bytes=0.0, flops=0.0))
            d["launches"] += 1
            d["seconds"] += s.elapsed_time(e) * 1e-3
            d["bytes"] += nbytes
            d["flops"] += fl
        return by

    def by_symbol(self):
        """{kernel instantiation as rocprofv3 names it (MFMA kernels) or wrapper name: dict(launches, seconds, flops, bytes,
        name)} - same shape as GemmProfile.by_symbol()."""
        torch.cuda.synchronize()
        by = {}
        for name, nbytes, s, e, fl, sym, _ in self.records:
            d = by.setdefault(sym, dict(launches=0, seconds=0.0, flops=0.0, bytes=0.0, name=name))
            d["launches"] += 1
            d["seconds"] += s.elapsed_time(e) * 1e-3
            d["flops"] += fl
            d["bytes"] += nbytes
        return by


class _hbm_op:
    """`with _hbm_op(name, bytes):` around ONE library call (no-op unless an OpProfile is active).  flops > 0: an MFMA
    kernel - its algorithmic FLOPs and the instantiation the call launched (vx_last_kernel) are recorded too."""

    def __init__(self, name, nbytes, flops=0.0):
        self.prof, self.name, self.nbytes, self.flops = OpProfile.active, name, nbytes, flops

    def __enter__(self):
        if self.prof is not None:
            self.s, self.e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *a):
        if self.prof is not None:
            self.e.record()
            sym = _lib.vx_last_kernel().decode() if self.flops else self.name
            self.prof.records.append((self.name, float(self.nbytes), self.s, self.e, float(self.flops), sym,
                                      PROFILE_WEIGHT[0]))


_log = logging.getLogger("v_express_amd")
# Which implementation each block of each UNet level took, recorded the first time the decision is made for a geometry
# (a refused one-launch block used to fall back silently): {(block, geometry...): "path (reason)"}.
# A block that HAS a one-launch form at its width but was refused by the geometry (f = 24, hw % 8 != 0, ...) is a
# performance cliff and is logged as a WARNING once; everything else at INFO (logger "v_express_amd").
BLOCK_PATHS = {}


def _note_path(block, geom, fused, path, reason="", cliff=False):
    key = (block,) + tuple(geom)
    text = path + (f" ({reason})" if reason else "")
    if BLOCK_PATHS.get(key) != text:
        BLOCK_PATHS[key] = text
        (_log.warning if cliff and not fused else _log.info)("%s %s: %s", block, dict(geom), text)
    return fused


def block_paths():
    """{"block geometry": "path"} of every decision taken so far (bench.py prints it; tests read it)."""
    return {f"{k[0]} " + " ".join(f"{a}={b}" for a, b in k[1:]): v for k, v in BLOCK_PATHS.items()}


_WARNED = set()


def _warn_once(msg):
    if msg not in _WARNED:
        _WARNED.add(msg)
        import warnings
        warnings.warn(msg, stacklevel=3)


def _tile_key(p):
    return _lib.vx_gemm_config_name(C.byref(p)).decode()


def _launch_gemm(p, what):
    prof = GemmProfile.active
    if prof is None:
        L.check(_lib.vx_gemm(C.byref(p), _stream()), what)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    L.check(_lib.vx_gemm(C.byref(p), _stream()), what)
    e.record()
    prof.records.append((s, e, 2.0 * p.m * p.n * p.k, _tile_key(p), (p.m, p.n, p.k),     # fp8: k incl. the zero padding
                         _lib.vx_gemm_last_kernel().decode(), PROFILE_WEIGHT[0]))
    # algorithmic bytes: every input row once (c1 + c2 channels), the weights once, the output once, residual once
    rows_in = p.nb * p.h_in * p.w_in
    n_out = p.n // 2 if p.epi == L.VX_EPI_GEGLU else p.n
    nbytes = 2.0 * (rows_in * (p.c1 + p.c2) + p.n * p.k + p.m * n_out * (2 if p.out_f32 else 1) +
                    (p.m * p.n if p.residual else 0))
    prof.bytes_of[id(s)] = nbytes
    skey = (p.m, p.n, p.k, prof.records[-1][3])
    prof.shape_bytes[skey] = prof.shape_bytes.get(skey, 0.0) + nbytes


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk_bf16(t, name):
    if t.dtype != L.ELEM[0] or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA {L.ELEM[0]} tensor (the element type in force), got {t.dtype} on {t.device}")


def _row_stride(t):
    """Row stride (elements) of a 2-D / [.., C] view whose last dim is contiguous."""
    if t.stride(-1) != 1:
        raise ValueError("last dimension must be contiguous")
    t2 = t if t.dim() == 2 else t.reshape(-1, t.shape[-1]) if t.is_contiguous() else None
    if t2 is None:
        raise ValueError("activation view must be 2-D or contiguous")
    return t2.stride(0), t2.shape[0]


class ConvGeom:
    """Geometry of an implicit-GEMM convolution over NHWC frames."""

    def __init__(self, nb, h_in, w_in, kh=1, kw=1, stride=1, pad=0, upsample=0, pad_end=0, out_hw=None):
        """pad: zero padding before AND after each spatial axis; pad_end: extra zero padding after only
        (diffusers Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) + conv stride 2 -> pad=0, pad_end=1).  The
        kernel's gather zero-fills every tap that falls outside the stored image, so pad_end only changes the
        output size."""
        self.nb, self.h_in, self.w_in = nb, h_in, w_in
        self.kh, self.kw, self.stride, self.pad, self.upsample = kh, kw, stride, pad, upsample
        he, we = h_in << upsample, w_in << upsample
        self.h_out = (he + 2 * pad + pad_end - kh) // stride + 1
        self.w_out = (we + 2 * pad + pad_end - kw) // stride + 1
        if out_hw is not None:
            # a sub-window of the valid outputs (the caller advances the A pointer to the window's first input pixel:
            # ops.gemm(a_pixel_offset=...)); it must lie inside the stored image
            if out_hw[0] > self.h_out or out_hw[1] > self.w_out or pad or upsample or pad_end:
                raise ValueError("out_hw must be a sub-window of a pad-0 convolution's outputs")
            self.h_out, self.w_out = out_hw

    @property
    def m(self):
        return self.nb * self.h_out * self.w_out


FP8_PROJ = [False]


class fp8_projections:
    """`with ops.fp8_projections(True):` - the attention q/k/v/out projections of the blocks run on the fp8 GEMM
    (BASELINE.json configs[4]); set by UNet3DConditionModel.fp8_projections around its forward."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self.prev = FP8_PROJ[0]
        FP8_PROJ[0] = self.on

    def __exit__(self, *a):
        FP8_PROJ[0] = self.prev


def proj_layernorm(x, gamma, beta, eps=1e-5, **kw):
    """LayerNorm in front of an attention projection: Fp8Rows under fp8_projections, bf16 otherwise."""
    return layernorm_fp8(x, gamma, beta, eps, **kw) if FP8_PROJ[0] else layernorm(x, gamma, beta, eps, **kw)


def proj_input(a):
    """An attention output about to enter its out-projection."""
    return quantize_fp8(a) if FP8_PROJ[0] else a


def proj_weight(a, w):
    """The weight operand matching the activation operand."""
    return fp8_weight(w) if isinstance(a, Fp8Rows) else w


class Fp8Rows:
    """Row-quantised activations for the fp8 projection GEMMs: q uint8 [rows, Kp] (OCP e4m3 bytes, K zero-padded to a
    multiple of 128), scale float32 [rows] (true value = q * scale), k = the unpadded width."""

    def __init__(self, q, scale, k):
        self.q, self.scale, self.k = q, scale, k

    @property
    def shape(self):
        return (self.q.shape[0], self.k)

    @property
    def device(self):
        return self.q.device


class Fp8Weight:
    """Per-output-row quantised weight: w8 uint8 [N, Kp] e4m3, scale float32 [N]."""

    def __init__(self, w8, scale, k):
        self.w8, self.scale, self.k = w8, scale, k

    @property
    def shape(self):
        return (self.w8.shape[0], self.k)


def pad128(k):
    return (k + 127) // 128 * 128


_FP8_W = {}


def fp8_weight(w):
    """bf16 [N, K] weight -> Fp8Weight (scale[n] = max|w[n, :]| / 448, round-to-nearest-even e4m3), cached per tensor."""
    key = (w.data_ptr(), tuple(w.shape))
    hit = _FP8_W.get(key)
    if hit is None:
        n, k = w.shape
        wf = w.float()
        sc = wf.abs().amax(dim=1).clamp_min(1e-30) / 448.0
        w8 = torch.zeros((n, pad128(k)), device=w.device, dtype=torch.uint8)
        w8[:, :k] = (wf / sc[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
        hit = _FP8_W[key] = (w, Fp8Weight(w8, sc.contiguous(), k))        # keep `w` alive: the key is its address
    return hit[1]


# the folded-LayerNorm GEMMs exist for the FAST addressing path only: the VX_GEMM_NOFAST A/B knob turns the fold off too
LN_FOLD = [os.environ.get("VX_LN_FOLD", "1") != "0" and os.environ.get("VX_GEMM_NOFAST") is None]
# VX_FUSED_STATS=0 (A/B knob): the row statistics a GEMM is asked for come from a separate vx_row_stats pass
FUSED_STATS = [os.environ.get("VX_FUSED_STATS", "1") != "0"]


# False (tests only): an all-zero-audio batch item gets its constant attn2 term from a vx_add_row_bias pass behind the
# reference attention - the same rounding placement as the fully computed block - instead of riding in the attn1
# out-projection's epilogue (blocks._spatial_transformer_read)
FOLD_ZERO_AUDIO = [True]
# VX_FF_SLAB_MB (A/B knob, default 0 = whole launches): see blocks._feed_forward
FF_SLAB_BYTES = [int(float(os.environ.get("VX_FF_SLAB_MB", "0")) * (1 << 20))]
# VX_GN_FUSED=0 (A/B knob): GroupNorm statistics always come from vx_groupnorm's own read pass
GN_FUSED = [os.environ.get("VX_GN_FUSED", "1") != "0"]


class GnStats:
    """GroupNorm partial sums of a tensor, written by the GEMM that produced it (vx_gemm_params.gn_ws): `ws` float32
    [frames, slabs, groups, 2] in vx_groupnorm's workspace layout.  Travels as the `_vx_gn` attribute of the tensor
    object (`keep_gn` carries it across a .view())."""

    def __init__(self, ws, slabs, groups, frames, hw, c):
        self.ws, self.slabs, self.groups, self.frames, self.hw, self.c = ws, slabs, groups, frames, hw, c

    def fits(self, frames, hw, groups, c):
        return (self.frames, self.hw, self.groups, self.c) == (frames, hw, groups, c)


def gn_of(x):
    """The GnStats its producer attached to `x`, or None."""
    return getattr(x, "_vx_gn", None) if GN_FUSED[0] else None


def _set_gn(t, gst=None):
    """`t` was just (re)written by a launch: attach the GroupNorm partial sums that launch produced, or - when it produced
    none - drop whatever an earlier producer left on the same tensor object (a reused `out=` buffer must never carry
    statistics of its previous contents into `groupnorm()`)."""
    if gst is not None:
        t._vx_gn = gst
    elif getattr(t, "_vx_gn", None) is not None:
        del t._vx_gn


def keep_gn(view, src):
    """`view` is a reshape of `src`: carry the producer's GroupNorm statistics over to the new tensor object."""
    st = getattr(src, "_vx_gn", None)
    if st is not None:
        view._vx_gn = st
    return view


def row_stats(x, eps=1e-5, out=None):
    """float32 [rows, 2] = (mean, rstd) of every row: the statistics of a LayerNorm folded into its consumer GEMM
    (`gemm(..., ln=(stats, colsum))`, weights.fold_layernorm)."""
    _chk_bf16(x, "x")
    ldx, rows = _row_stride(x)
    if out is None:
        out = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    elif out.dtype != torch.float32 or not out.is_contiguous() or tuple(out.shape) not in ((rows, 2), (rows, 4)):
        raise ValueError("row_stats: out must be a contiguous float32 [rows, 2] (mean, rstd) or [rows, 4] (two-part sums) tensor")
    with _hbm_op("row_stats", rows * (2 * x.shape[-1] + 8)):
        if out.shape[1] == 4:
            L.check(_lib.vx_row_stats_parts(_ptr(x), ldx, rows, x.shape[-1], _ptr(out), _stream()), "vx_row_stats_parts")
        else:
            L.check(_lib.vx_row_stats(_ptr(x), ldx, rows, x.shape[-1], float(eps), _ptr(out), _stream()), "vx_row_stats")
    return out


# Row statistics in TWO parts (vx_gemm_params.row_stats_parts / ln_stats_parts): at the widths listed here a statistics
# buffer is float32 [rows, 4] = (sum, sum of squares) of each half row instead of [rows, 2] = (mean, rstd) - the format is
# the buffer's own width, so slices carry it.  At 640 channels (the 32x32 level) one tile of the persistent kernel holds
# half a row: each half comes out of a tile's epilogue and the vx_row_stats pass behind the GEMM disappears.
# VX_STATS_PARTS=0 restores the (mean, rstd) format everywhere (A/B knob).
STATS_PARTS_WIDTHS = {640} if os.environ.get("VX_STATS_PARTS", "1") != "0" else set()


def stats_buffer(rows, c, device):
    """The row-statistics buffer of a residual stream of width c: [rows, 4] (two-part sums) or [rows, 2] (mean, rstd)."""
    two = c in STATS_PARTS_WIDTHS and FUSED_STATS[0] and LN_FOLD[0] and not FP8_PROJ[0]
    return torch.empty((rows, 4 if two else 2), device=device, dtype=torch.float32)


def _set_ln(p, ln, eps=1e-5):
    if ln is None:
        return
    stats, colsum = ln
    if stats.dtype != torch.float32 or colsum.dtype != torch.float32 or not stats.is_contiguous() or \
            tuple(stats.shape) not in ((p.m, 2), (p.m, 4)) or colsum.numel() != p.n:
        raise ValueError("ln=(stats [m, 2] or [m, 4] float32, colsum [n] float32) expected")
    p.ln_stats, p.ln_colsum = stats.data_ptr(), colsum.data_ptr()
    if stats.shape[1] == 4:
        # two-part sums: the persistent kernel finishes them in its epilogue; any other launch gets them finished first
        p.ln_stats_parts, p.ln_eps = 2, float(eps)
        if p.k != 640 or not _lib.vx_gemm_config_name(C.byref(p)).decode().startswith("gemm_ring"):
            # (5 such launches of ~5 us per DDIM step at 512x512 - profiles/r04y_trace_summary.txt: the V^T halves of the
            # 32x32-level QKV projections; every other consumer at that level runs on the persistent kernel - so the
            # finished statistics are not cached across consumers: each statistics version has ONE consumer of this kind)
            fin = torch.empty((p.m, 2), device=stats.device, dtype=torch.float32)
            L.check(_lib.vx_row_stats_finalize(_ptr(stats), p.m, p.k, float(eps), _ptr(fin), _stream()),
                    "vx_row_stats_finalize")
            p.ln_stats, p.ln_stats_parts = fin.data_ptr(), 0
            p._keep_ln = fin            # alive until the launch is enqueued (same stream: the allocator orders the reuse)


def layernorm_fp8(x, gamma, beta, eps=1e-5, *, add=None, add_rows_per_entry=1, add_entries=1):
    """LayerNorm (gamma=None: no normalisation, plain row quantisation) -> Fp8Rows."""
    _chk_bf16(x, "x")
    ldx, rows = _row_stride(x)
    c = x.shape[-1]
    kp = pad128(c)
    q = torch.empty((rows, kp), device=x.device, dtype=torch.uint8)
    sc = torch.empty((rows,), device=x.device, dtype=torch.float32)
    L.check(_lib.vx_layernorm_fp8(_ptr(x), ldx, rows, c, float(eps), _ptr(gamma), _ptr(beta), _ptr(add),
                                  add_rows_per_entry, add_entries, _ptr(q), kp, _ptr(sc), _stream()),
            "vx_layernorm_fp8")
    return Fp8Rows(q, sc, c)


def quantize_fp8(x):
    return layernorm_fp8(x, None, None)


def _base_params(a, w, geom, a2=None):
    """a: [rows, C1] view (row stride allowed); a2 likewise; w: [N, K] bf16 contiguous.
    a: Fp8Rows + w: Fp8Weight -> the fp8 GEMM (plain linears)."""
    if isinstance(a, Fp8Rows):
        if not isinstance(w, Fp8Weight) or a2 is not None or geom is not None:
            raise TypeError("fp8 activations need an Fp8Weight and a plain linear (no conv geometry, one source)")
        if a.q.shape[1] != w.w8.shape[1]:
            raise ValueError(f"fp8 K mismatch: {a.q.shape[1]} vs {w.w8.shape[1]}")
        p = L.GemmParams()
        rows, kp = a.q.shape
        p.a, p.c1, p.lda1 = a.q.data_ptr(), kp, a.q.stride(0)
        p.a2, p.c2, p.lda2 = None, 0, 0
        geom = ConvGeom(1, rows, 1)
        p.nb, p.h_in, p.w_in = 1, rows, 1
        p.kh, p.kw, p.stride, p.pad, p.upsample = 1, 1, 1, 0, 0
        p.h_out, p.w_out = rows, 1
        p.w, p.n, p.k, p.m = w.w8.data_ptr(), w.w8.shape[0], kp, rows
        p.alpha = 1.0
        p.a_fp8, p.a_scale, p.w_scale = 1, a.scale.data_ptr(), w.scale.data_ptr()
        return p, geom
    _chk_bf16(a, "a")
    _chk_bf16(w, "w")
    p = L.GemmParams()
    lda1, rows = _row_stride(a)
    p.a, p.c1, p.lda1 = a.data_ptr(), a.shape[-1], lda1
    if a2 is not None:
        _chk_bf16(a2, "a2")
        lda2, rows2 = _row_stride(a2)
        if rows2 != rows:
            raise ValueError("a/a2 row mismatch")
        p.a2, p.c2, p.lda2 = a2.data_ptr(), a2.shape[-1], lda2
    else:
        p.a2, p.c2, p.lda2 = None, 0, 0
    if geom is None:
        geom = ConvGeom(1, rows, 1)
    if rows != geom.nb * geom.h_in * geom.w_in:
        raise ValueError(f"activation rows {rows} != nb*h*w {geom.nb * geom.h_in * geom.w_in}")
    p.nb, p.h_in, p.w_in = geom.nb, geom.h_in, geom.w_in
    p.kh, p.kw, p.stride, p.pad, p.upsample = geom.kh, geom.kw, geom.stride, geom.pad, geom.upsample
    p.h_out, p.w_out = geom.h_out, geom.w_out
    if not w.is_contiguous() or w.dim() != 2:
        raise ValueError("weight must be a contiguous [N, K] matrix")
    p.w, p.n, p.k, p.m = w.data_ptr(), w.shape[0], w.shape[1], geom.m
    p.alpha = 1.0
    return p, geom


_SPLITK_WS = {}
# split-K policy knobs (A/B measurements): rows of the tile the launch will get and the block count to aim for
_SPLITK_ROWS = int(os.environ.get("VX_SPLITK_ROWS", "128"))
_SPLITK_TARGET = int(os.environ.get("VX_SPLITK_TARGET", "512"))
_FRAME_ROWS = [None]


def _stream_key(device):
    """Persistent scratch buffers (split-K workspace, zero-bordered GroupNorm images) are reused stream-ordered: the
    consumer of a buffer is enqueued before its next producer ON THE SAME STREAM.  Keying them by the current stream
    keeps two streams (or two models driven from two streams) from sharing one; work on different streams that must
    share data is the caller's to order with events, as for any torch tensor."""
    return torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0


def clear_caches():
    """Free the persistent scratch buffers (they are re-created on demand).  Call between clips of different geometry
    to return the memory: one zero-bordered image exists per distinct (frames, H, W, C) seen so far."""
    _SPLITK_WS.clear()
    _PADDED.clear()
    _FP8_W.clear()
    _FF_PACKED.clear()
    _TB_PACKED.clear()
    _COOP_WS.clear()
    _COOP_OK.clear()


_ITEMS = [None]


class frame_rows:
    """`with ops.frame_rows(hw, items=b):` tells the GEMM wrappers how many token rows one frame holds and how many
    independent batch items (CFG halves) share the launch.  Only the kernel-selection policies read it (split-K factor,
    ring-kernel hint): they must be functions of per-item facts, never of how many items are batched, so that a CFG
    half computed alone (on another GPU) is bit-identical to its rows in the batched call."""

    def __init__(self, hw, items=None):
        self.hw, self.items = hw, items

    def __enter__(self):
        self.prev = (_FRAME_ROWS[0], _ITEMS[0])
        _FRAME_ROWS[0], _ITEMS[0] = self.hw, self.items

    def __exit__(self, *a):
        _FRAME_ROWS[0], _ITEMS[0] = self.prev


# A/B knobs of the kernel choice.  They live HERE (ABI 14): the library has no process-wide mode any more, every vx_gemm call
# carries its own choice in vx_gemm_params.ring_hint, so two pipelines in one process cannot race on a switch.
#   RING_MODE (VX_GEMM_RING): 2 = the persistent ring-staged kernel for every eligible problem (product), 1 = only K <= 1280,
#   0 = never (results then differ only by fp32 summation order: the ring kernel walks conv taps innermost)
#   FP8_RING (VX_FP8_RING=1): fp8 operands on the persistent kernel (ring_hint = 3; measures slower than the classic fp8 tiles)
RING_MODE = [int(os.environ.get("VX_GEMM_RING", "2")) if os.environ.get("VX_GEMM_RING", "2") in ("0", "1", "2") else 2]
FP8_RING = [os.environ.get("VX_FP8_RING") == "1"]


def _ring_off(k):
    return RING_MODE[0] == 0 or (RING_MODE[0] == 1 and k > 1280)


def _ring_hint(p):
    """vx_gemm_params.ring_hint from batch-independent facts: rows of ONE batch item (a 16-frame CFG half) and N.
    The ring kernel is chosen when a nominal CFG-pair launch (2 items) would have >= 192 tiles of 256 x 320 and an
    item is a whole number of 256-row tiles; without the frame_rows context the library decides by the launch size."""
    if _ring_off(p.k):
        return -1
    items = _ITEMS[0]
    if items is None or items <= 0 or p.m % items:
        return 0
    rows_item = p.m // items
    if rows_item % 256 or p.n % 320:
        return -1
    return 1 if (2 * rows_item // 256) * (p.n // 320) >= 192 else -1


QK_RING = [os.environ.get("VX_QK_RING", "1") != "0"]


def qk_on_ring(m, c):
    """Whether the Q | K half of a fused, LayerNorm-folded QKV projection ([m, c] -> [m, 2c]) would run on the persistent
    ring kernel (same batch-independent rule as `_ring_hint`): then blocks._self_attention issues it there and leaves
    only V^T to the classic SPLIT epilogue."""
    items = _ITEMS[0]
    if not QK_RING[0] or items is None or items <= 0 or m % items or (2 * c) % 320 or c % 64:
        return False
    rows_item = m // items
    return rows_item % 256 == 0 and (2 * rows_item // 256) * (2 * c // 320) >= 192 and not _ring_off(c)


def _splitk(p, geom, device, plain):
    """Split-K factor for the 8x8-level problems (M = frames * 64 rows: half the CUs idle otherwise).  A function of
    the per-frame geometry, N and K only - never of the number of frames - so a CFG half or a window computed alone
    sums in the same order as the batched call (bit-identical)."""
    nk = -(-p.k // 64)
    hw = _FRAME_ROWS[0] if plain else geom.h_out * geom.w_out
    # K < 2560 (the 1x1 / linear layers of the 8x8 level): one launch on the 64 x 160 tile (vx_gemm picks it when the
    # 128-row tiling leaves CUs idle) beats split-K + reduce: 21.1 vs 27.4 us at 2048 x 1280 x 1280
    # (profiles/r02b_gemm_small.txt); the long-K 3x3 convs keep the split (77.7 vs 122.7 us)
    if hw is None or hw > 64 or p.k < 2560:
        return
    tiles = (2048 // _SPLITK_ROWS) * -(-p.n // 160)    # nominal 32-frame launch (2048 rows at the 8x8 level)
    s = min(-(-_SPLITK_TARGET // tiles), 8, nk // 4)
    if s < 2:
        return
    nbytes = int(_lib.vx_gemm_splitk_ws_bytes(p.m, p.n, s))
    key = (device, _stream_key(device), nbytes)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = _SPLITK_WS[key] = torch.empty(nbytes, device=device, dtype=torch.uint8)
    p.splitk, p.splitk_ws = s, ws.data_ptr()


# the 16x16-level launches (256 rows per frame, N = 1280: 128 tiles of 256 x 320 for a CFG pair) on
# the persistent kernel with a cooperative two-way K split (vx_gemm_params.ring_hint = 2).  VX_RING_COOP=0 keeps them on
# the 128 x 160 tiles (A/B knob); COOP_MIN_K: shortest K that takes the split.
RING_COOP = [os.environ.get("VX_RING_COOP", "1") != "0"]
COOP_MIN_K = [int(os.environ.get("VX_RING_COOP_MIN_K", "8192"))]
_COOP_WS = {}            # (device, stream, bytes) -> [zeroed workspace, epoch of its last launch]
_COOP_OK = {}            # epilogue / geometry signature -> vx_gemm_ring_coop_ok (asked of the library once per signature)
_COOP_EPOCH_MAX = (1 << 27) - 1


def _coop_signature(p, rows_item):
    """Every fact the decision reads (launches that share (rows_item, n, k) but differ in the epilogue decided
    differently under one log key): the shape, the K chunking, the epilogue's operands and strides."""
    return (rows_item, p.m, p.n, p.k, p.c1, p.c2, p.kh, p.kw, p.stride, p.lda1, p.lda2, p.ldc, p.ldr, bool(p.residual),
            bool(p.ln_stats), bool(p.row_stats_out), p.row_stats_parts, p.w_group_rows, bool(p.rowbias), p.rowbias_ld,
            p.rows_per_group, p.epi, p.out_f32, p.act)


def ring_coop_applies(p):
    """Whether the launch p takes the cooperative split - a function of batch-independent facts only (rows of ONE item, N,
    K, the epilogue; never of how many items share the launch), so a CFG half computed alone sums in the same order.
    Only where `_ring_hint` said "not the ring kernel" because a CFG pair has too few tiles: rows of an item a multiple of
    256, 96 <= tiles of the nominal pair < 192, K >= COOP_MIN_K; the kernel's own limits are asked of the library."""
    items = _ITEMS[0]
    if items is None or items <= 0 or p.m % items or _ring_off(p.k):
        return False
    rows_item = p.m // items
    if rows_item % 256 or p.n % 320:
        return False
    tiles_pair = (2 * rows_item // 256) * (p.n // 320)
    if not 96 <= tiles_pair < 192:
        return False
    sig = _coop_signature(p, rows_item)
    lib_ok = _COOP_OK.get(sig)
    if lib_ok is None and RING_COOP[0] and p.k >= COOP_MIN_K[0]:
        lib_ok = _COOP_OK[sig] = bool(_lib.vx_gemm_ring_coop_ok(C.byref(p)))
    why = ("VX_RING_COOP=0" if not RING_COOP[0] else f"K below {COOP_MIN_K[0]}: break-even" if p.k < COOP_MIN_K[0] else
           "" if lib_ok else "the kernel's limits (epilogue / odd number of 64-channel chunks)")
    # logged once per deciding signature like the one-launch blocks' decisions (bench.py: block_paths)
    epi = ("+res" if p.residual else "") + ("+ln" if p.ln_stats else "") + ("+rowstats" if p.row_stats_out else "")
    return _note_path("gemm_too_few_tiles", (("rows_item", rows_item), ("n", p.n), ("k", p.k), ("chunks", (p.c1 + p.c2) // 64),
                                             ("epilogue", epi or "plain")), not why,
                      "persistent kernel, cooperative two-way K split" if not why else "128 x 160 tiles", why)


def _ring_coop(p, device):
    """Switch p over to the cooperative split (ring_hint = 2, splitk = 2, the workspace and its next epoch) when
    `ring_coop_applies`.  The rendezvous flags carry the launch's EPOCH (vx_gemm_params.coop_epoch, ABI 14) and are never
    reset: a launch that lost a partner (bounded poll) or was aborted leaves nothing a later launch can mistake for its own
    (with reset-to-zero flags one stale word poisoned every later launch of the process)."""
    if not ring_coop_applies(p):
        return False
    nbytes = int(_lib.vx_gemm_splitk_ws_bytes(p.m, p.n, 2))
    key = (device, _stream_key(device), nbytes)
    ent = _COOP_WS.get(key)
    if ent is None or ent[1] >= _COOP_EPOCH_MAX:
        ent = _COOP_WS[key] = [torch.zeros(nbytes, device=device, dtype=torch.uint8), 0]   # epoch 0 = "never written"
    ent[1] += 1
    p.splitk, p.splitk_ws, p.ring_hint, p.coop_epoch = 2, ent[0].data_ptr(), 2, ent[1]
    return True


def gemm(a, w, bias=None, *, geom=None, a2=None, residual=None, alpha=1.0, act=L.VX_ACT_NONE, rowbias=None,
         rows_per_group=0, out=None, out_f32=False, ln=None, stats_out=None, stats_eps=1e-5, w_group_rows=0, gn=None,
         a_pixel_offset=0):
    """out[m, n] = residual + alpha * act(sum_k A[m,k] W[n,k] + bias[n] + rowbias[m // rows_per_group, n]).
    ln=(stats, colsum): a LayerNorm folded into this GEMM - the sum is replaced by rstd[m] * (sum - mean[m] * colsum[n])
    with `w`, `bias` the folded weight / bias of weights.fold_layernorm and `a` the un-normalised rows.
    stats_out: float32 [m, 2] (contiguous) that receives (mean, rstd) of every STORED output row - the `ln` statistics
    of the next GEMM (produced by the epilogue itself at the 64x64 level, by vx_row_stats inside vx_gemm otherwise).
    w_group_rows > 0: `w` is [m // w_group_rows, N, K]; output rows of group g use w[g] (groupnorm_fold_linear).
    gn=(groups, hw): the next reader of `out` is a GroupNorm with `groups` groups over frames of `hw` rows - when the
    launch can (vx_gemm_gn_slabs), its epilogue also writes that GroupNorm's partial sums and the returned tensor carries
    them (`gn_of(out)`), so `groupnorm(out, ...)` skips its statistics pass; otherwise nothing changes."""
    plain = geom is None
    if w_group_rows:
        if w.dim() != 3 or not w.is_contiguous():
            raise ValueError("grouped weights must be a contiguous [groups, N, K] tensor")
        n_groups = w.shape[0]
        w = w.view(-1, w.shape[-1])
    p, geom = _base_params(a, w, geom, a2)
    if a_pixel_offset:
        # the convolution window starts `a_pixel_offset` pixels into every frame (ConvGeom(out_hw=...)): its last tap must
        # stay inside the frame
        last = a_pixel_offset + ((geom.h_out - 1) * geom.stride + geom.kh - 1) * geom.w_in + (geom.w_out - 1) * geom.stride + geom.kw - 1
        if a2 is not None or a_pixel_offset < 0 or last >= geom.h_in * geom.w_in:
            raise ValueError("a_pixel_offset: the shifted window leaves the frame")
        p.a = a.data_ptr() + a_pixel_offset * p.lda1 * a.element_size()
    if w_group_rows:
        if p.m != n_groups * w_group_rows:
            raise ValueError(f"{n_groups} weight groups of {w_group_rows} rows do not cover m={p.m}")
        p.n //= n_groups
        p.w_group_rows = w_group_rows
    n = p.n
    if out is None:
        out = torch.empty((geom.m, n), device=a.device, dtype=torch.float32 if out_f32 else L.ELEM[0])
    ldc, orows = _row_stride(out)
    if orows != geom.m or out.shape[-1] != n:
        raise ValueError("bad output shape")
    p.epi, p.act, p.alpha = L.VX_EPI_STORE, act, float(alpha)
    p.bias = bias.data_ptr() if bias is not None else None
    if rowbias is not None:
        if rowbias.dtype != torch.float32 or rowbias.stride(-1) != 1:
            raise TypeError("rowbias must be float32 with contiguous columns")
        p.rowbias, p.rowbias_ld, p.rows_per_group = rowbias.data_ptr(), rowbias.stride(0), rows_per_group
    if residual is not None:
        _chk_bf16(residual, "residual")
        p.residual, p.ldr = residual.data_ptr(), _row_stride(residual)[0]
    p.out, p.ldc, p.out_f32 = out.data_ptr(), ldc, int(out_f32)
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("bias must be float32")
    if not p.a_fp8:
        if ln is None:
            _splitk(p, geom, a.device, plain)
        p.ring_hint = _ring_hint(p)
    elif FP8_RING[0]:
        p.ring_hint = 3
    _set_ln(p, ln)
    if stats_out is not None:
        if stats_out.dtype != torch.float32 or not stats_out.is_contiguous() or out_f32 or \
                tuple(stats_out.shape) not in ((p.m, 2), (p.m, 4)):
            raise ValueError("stats_out must be a contiguous float32 [m, 2] / [m, 4] tensor (bf16 output only)")
        if FUSED_STATS[0]:
            p.row_stats_out, p.row_stats_eps = stats_out.data_ptr(), float(stats_eps)
            p.row_stats_parts = 2 if stats_out.shape[1] == 4 else 0
            if p.row_stats_parts and p.ring_hint == 0:
                # which kernel sums the half rows (the persistent kernel's epilogue or vx_row_stats_parts - different
                # fp32 summation orders) then depends on the launch size: batch invariance of the two-part statistics
                # holds only under `with ops.frame_rows(hw, items=...)`, as every call site in blocks.py has it
                _warn_once("two-part row statistics requested outside ops.frame_rows(items=...): the kernel that sums "
                           "them is chosen by the launch size, so the low bits depend on the batch")
    if not p.a_fp8 and p.ring_hint == -1 and p.splitk <= 1 and not out_f32:
        _ring_coop(p, a.device)
    gst = None
    if gn is not None and GN_FUSED[0] and not p.a_fp8:
        groups, hw = gn
        p.gn_groups, p.gn_hw = int(groups), int(hw)
        slabs = int(_lib.vx_gemm_gn_slabs(C.byref(p)))
        if slabs > 0:
            frames = p.m // hw
            gst = GnStats(torch.empty((frames, slabs, groups, 2), device=a.device, dtype=torch.float32), slabs, groups,
                          frames, hw, n)
            p.gn_ws = gst.ws.data_ptr()
    _launch_gemm(p, "vx_gemm")
    _set_gn(out, gst)
    if stats_out is not None and not FUSED_STATS[0]:
        row_stats(out, stats_eps, out=stats_out)       # A/B arm: the separate read pass of round 2
    return out


def geglu(a, w_interleaved, bias_interleaved, out=None, ln=None):
    """FeedForward first half: h * gelu(g) with value/gate weight rows interleaved in blocks of 8 (ln: as in gemm)."""
    p, geom = _base_params(a, w_interleaved, None)
    if out is None:
        out = torch.empty((geom.m, p.n // 2), device=a.device, dtype=L.ELEM[0])
    p.epi = L.VX_EPI_GEGLU
    p.bias = bias_interleaved.data_ptr() if bias_interleaved is not None else None
    p.out, p.ldc = out.data_ptr(), _row_stride(out)[0]
    p.ring_hint = _ring_hint(p)
    _set_ln(p, ln)
    _launch_gemm(p, "vx_gemm(geglu)")
    _set_gn(out)
    return out


# The 64x64-level feed-forward (C = 320) as ONE launch (vx_ff_fused, round 4): bit-identical to the two vx_gemm launches
# (same arithmetic chain in the same order), 345 vs 405 us per FF, +0.8 % on the whole path (profiles/r04d_*, r04e_*).
# VX_FF_FUSED=0 restores the two launches (A/B knob).
FF_FUSED = [os.environ.get("VX_FF_FUSED", "1") != "0"]
_FF_PACKED = {}


def ff_fused_applies(m, c, hidden):
    why = ("VX_FF_FUSED=0" if not FF_FUSED[0] else "fp8 projection mode" if FP8_PROJ[0] else
           "the one-launch kernel holds a tile's rows in registers: C = 320 only" if (c != 320 or hidden != 1280) else
           f"rows {m} not a multiple of 128" if m % 128 else "")
    return _note_path("feed_forward", (("c", c), ("hidden", hidden), ("rows%128", m % 128)), not why,
                      "one launch (vx_ff_fused)" if not why else "two launches (GEGLU GEMM + output GEMM)", why,
                      cliff=(c == 320 and hidden == 1280))


def ff_fused(h, w1_folded, b1, colsum, stats, w2, b2):
    """h += (value * gelu(gate)) w2^T + b2 with [value | gate] = LN(h) w1^T + b1, LayerNorm folded as in `geglu(ln=...)`;
    in place on h (rows are independent).  The two weights are re-tiled once per weight tensor (vx_ff_pack_weights)."""
    _chk_bf16(h, "h")
    ldx, m = _row_stride(h)
    c, hidden = h.shape[-1], w2.shape[1]
    if stats.dtype != torch.float32 or not stats.is_contiguous() or tuple(stats.shape) != (m, 2):
        raise ValueError("ff_fused: statistics must be a contiguous float32 [rows, 2] (mean, rstd) tensor")
    key = (w1_folded.data_ptr(), w2.data_ptr())
    hit = _FF_PACKED.get(key)
    if hit is None:
        w1t, w2t = torch.empty_like(w1_folded), torch.empty_like(w2)
        L.check(_lib.vx_ff_pack_weights(_ptr(w1_folded), _ptr(w2), _ptr(w1t), _ptr(w2t), c, hidden, _stream()),
                "vx_ff_pack_weights")
        hit = _FF_PACKED[key] = (w1_folded, w2, w1t, w2t)          # keep the sources alive: the key is their address
    p = L.FfParams()
    p.x, p.ldx, p.m, p.c, p.hidden = h.data_ptr(), ldx, m, c, hidden
    p.w1t, p.w2t = hit[2].data_ptr(), hit[3].data_ptr()
    p.bias1 = b1.data_ptr() if b1 is not None else None
    p.ln_colsum, p.ln_stats = colsum.data_ptr(), stats.data_ptr()
    p.bias2 = b2.data_ptr() if b2 is not None else None
    p.residual, p.ldr, p.out, p.ldo = h.data_ptr(), ldx, h.data_ptr(), ldx
    # reads the rows once, writes them once (bf16) + both weights; 2 m c (2 hidden) + 2 m hidden c FLOP
    with _hbm_op("ff_fused", 2 * (2 * m * c + 3 * hidden * c), flops=6.0 * m * c * hidden):
        L.check(_lib.vx_ff_fused(C.byref(p), _stream()), "vx_ff_fused")
    _set_gn(h)
    return h


# Temporal self-attention block of the 64x64 level in ONE launch (csrc/vx_tblock.hip): LayerNorm-folded QKV projection,
# attention over the window's frames (16, or - round 5 - the reference's default 24), out-projection and residual;
# VX_TB_FUSED=0 restores the three launches (A/B knob).
TB_FUSED = [os.environ.get("VX_TB_FUSED", "1") != "0"]
# window lengths the one-launch kernel is built for -> pixels per tile (a tile = TB_PIX[f] pixels x their f frames)
TB_FRAMES = (16, 24)
TB_PIX = {16: 8, 24: 4}
_TB_PACKED = {}


def tblock_fused_applies(c, heads, f, hw):
    why = ("VX_TB_FUSED=0" if not TB_FUSED[0] else "LayerNorm fold off" if not LN_FOLD[0] else
           "fp8 projection mode" if FP8_PROJ[0] else
           "the one-launch kernel holds a tile's rows in registers: C = 320, 8 heads only" if (c != 320 or heads != 8) else
           f"window of {f} frames: the one-launch kernel is built for f in {TB_FRAMES}" if f not in TB_FRAMES else
           f"{hw} pixels per frame not a multiple of the tile's {TB_PIX[f]}" if hw % TB_PIX[f] else "")
    return _note_path("temporal_attention", (("c", c), ("heads", heads), ("f", f), ("hw%tile", hw % TB_PIX.get(f, 8))),
                      not why, "one launch (vx_tblock_fused)" if not why else
                      "three launches (QKV GEMM + temporal attention + output GEMM)", why, cliff=(c == 320 and heads == 8))


def tblock_fused(h, wqkv_folded, bqkv, colsum, pe_rows, wo, bo, *, b, f, hw, heads, stats=None, stats_out=None, eps=1e-5):
    """h += to_out(attention over f of (LN(h) + pe) Wqkv^T + b), in place on h [(b f) hw, C] (a tile of the kernel is 8
    pixels x their 16 frames, or 4 pixels x 24 frames: rows of other pixels are independent).  stats: (mean, rstd) per row, or None - the kernel
    then takes them from the rows it holds.  stats_out: float32 [rows, 2] that receives (mean, rstd) of the rows written
    (for the next LayerNorm fold; may be the same tensor as stats).  Weights and tables are re-tiled once per layer
    (vx_tblock_pack)."""
    _chk_bf16(h, "h")
    ldx, m = _row_stride(h)
    c = h.shape[-1]
    if m != b * f * hw:
        raise ValueError("h rows != b*f*hw")
    for t_ in (stats, stats_out):
        if t_ is not None and (t_.dtype != torch.float32 or not t_.is_contiguous() or tuple(t_.shape) != (m, 2)):
            raise ValueError("tblock_fused: statistics must be contiguous float32 [rows, 2] (mean, rstd) tensors")
    key = (wqkv_folded.data_ptr(), wo.data_ptr(), pe_rows.data_ptr() if pe_rows is not None else 0, f)
    hit = _TB_PACKED.get(key)
    if hit is None:
        dev = h.device
        wqkv_t = torch.empty(int(_lib.vx_tblock_packed_bytes(f)) // 2, device=dev, dtype=L.ELEM[0])
        wo_t = torch.empty(204800 // 2, device=dev, dtype=L.ELEM[0])
        cs = torch.empty(1024, device=dev, dtype=torch.float32)
        L.check(_lib.vx_tblock_pack(_ptr(wqkv_folded), _ptr(bqkv) if bqkv is not None else None, _ptr(colsum),
                                    _ptr(pe_rows) if pe_rows is not None else None,
                                    pe_rows.stride(0) if pe_rows is not None else 0, _ptr(wo), _ptr(wqkv_t), _ptr(wo_t),
                                    _ptr(cs), c, heads, f, _stream()), "vx_tblock_pack")
        hit = _TB_PACKED[key] = (wqkv_folded, wo, pe_rows, wqkv_t, wo_t, cs)   # sources kept alive: the key is their address
    p = L.TBlockParams()
    p.x, p.ldx, p.b, p.f, p.hw, p.c, p.heads = h.data_ptr(), ldx, b, f, hw, c, heads
    p.wqkv_t, p.wo_t, p.colsum_p = (t.data_ptr() for t in hit[3:6])
    p.bias_o = bo.data_ptr() if bo is not None else None
    p.ln_stats = stats.data_ptr() if stats is not None else None
    p.stats_out = stats_out.data_ptr() if stats_out is not None else None
    p.ln_eps, p.scale = eps, (c // heads) ** -0.5
    # reads the rows once, writes them once (bf16); QKV + out projections 2 m c (3c + c), attention over f 4 m f c FLOP
    with _hbm_op("tblock_fused", 2 * m * c * 2, flops=8.0 * m * c * c + 4.0 * m * f * c):
        L.check(_lib.vx_tblock_fused(C.byref(p), _stream()), "vx_tblock_fused")
    _set_gn(h)
    return h


def vt_pitch(n):
    return (n + 7) // 8 * 8


def gemm_split(a, w, bias, parts, *, part_cols, seq_len=0, head_dim=0, geom=None, ln=None):
    """One GEMM whose column ranges go to different destinations (ln: as in gemm).
    parts: list of ("rows", tensor[m, part_cols]) or ("vt", tensor[seqs, heads, head_dim, pitch])."""
    p, geom = _base_params(a, w, geom)
    p.epi = L.VX_EPI_SPLIT
    _set_ln(p, ln)                      # (after the epilogue is known: two-part statistics ask which kernel will run)
    p.bias = bias.data_ptr() if bias is not None else None
    p.part_cols, p.n_parts = part_cols, len(parts)
    p.seq_len, p.head_dim = seq_len, head_dim
    for i, (kind, t) in enumerate(parts):
        _chk_bf16(t, "part")
        p.part_out[i] = t.data_ptr()
        if kind == "rows":
            p.part_kind[i] = L.VX_PART_ROWS
            p.part_ld[i] = _row_stride(t)[0]
        else:
            p.part_kind[i] = L.VX_PART_VT
            p.part_ld[i] = 0
            p.vt_pitch = t.shape[-1]
            if not t.is_contiguous():
                raise ValueError("V^T destination must be contiguous")
    _launch_gemm(p, "vx_gemm(split)")


def alloc_vt(seqs, heads, head_dim, n, device):
    """V^T buffer [seqs, heads, head_dim, pitch]; zero-filled when the pitch pads the key axis."""
    pitch = vt_pitch(n)
    if pitch != n:
        return torch.zeros((seqs, heads, head_dim, pitch), device=device, dtype=L.ELEM[0])
    return torch.empty((seqs, heads, head_dim, pitch), device=device, dtype=L.ELEM[0])


def _gn_slices(hw):
    """Spatial slices per frame for the GroupNorm statistics.  A function of hw ONLY: the partial-sum grouping
    (hence the fp32 rounding) must not depend on how many frames share the launch, so that a CFG half or a
    window computed on another GPU is bit-identical to the batched call."""
    return max(1, min(64, hw // 16))


_PADDED = {}


def padded_buffer(device, frames, H, W, c):
    """Persistent zero-bordered NHWC image [frames, (H+2)*(W+2), c] per shape.  Only `groupnorm(pad_hw=...)` writes
    into it (interior pixels only), so the border stays zero for the life of the process; the one buffer per shape
    is reused stream-ordered (the conv that reads it is enqueued before the next GroupNorm that refills it)."""
    key = (device, _stream_key(device), frames, H, W, c, L.ELEM[0])    # one image per element type (bf16 / half models in one process)
    buf = _PADDED.get(key)
    if buf is None:
        buf = torch.zeros((frames, (H + 2) * (W + 2), c), device=device, dtype=L.ELEM[0])
        _PADDED[key] = buf
    return buf


def groupnorm(x1, gamma, beta, *, frames, hw, groups, eps, silu, x2=None, out=None, pad_hw=None):
    """x1: [frames, hw, C1] (+ x2: [frames, hw, C2] channel-concatenated) -> [frames, hw, C1+C2];
    pad_hw=(H, W): -> the zero-bordered image [frames, (H+2)*(W+2), C] (see `padded_buffer`), to be convolved with
    ConvGeom(frames, H+2, W+2, 3, 3, 1, pad=0).
    When the GEMM that produced x1 left its GroupNorm partial sums on the tensor (`gn_of`), only the apply pass runs."""
    _chk_bf16(x1, "x1")
    if not x1.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise ValueError("groupnorm inputs must be contiguous")
    c1 = x1.shape[-1]
    c2 = x2.shape[-1] if x2 is not None else 0
    width, pad = hw, 0
    if pad_hw is not None:
        H, W = pad_hw
        if H * W != hw or out is not None:
            raise ValueError("pad_hw must factor hw and excludes `out`")
        out = padded_buffer(x1.device, frames, H, W, c1 + c2)
        width, pad = W, 1
    elif out is None:
        out = torch.empty((frames, hw, c1 + c2), device=x1.device, dtype=L.ELEM[0])
    slices = _gn_slices(hw)
    st = gn_of(x1) if x2 is None else None
    elems = frames * hw * (c1 + c2)
    if st is not None and st.fits(frames, hw, groups, c1):
        with _hbm_op("groupnorm_apply", 4 * elems):                     # read 2 B + write 2 B per element
            L.check(_lib.vx_groupnorm_apply(_ptr(x1), c1, None, 0, frames, hw, groups, float(eps), _ptr(gamma),
                                            _ptr(beta), int(silu), _ptr(out), _ptr(st.ws), st.slabs, slices, width, pad,
                                            _stream()), "vx_groupnorm_apply")
        return out
    ws = torch.empty(int(_lib.vx_groupnorm_ws_floats(frames, slices, groups)), device=x1.device,
                     dtype=torch.float32)
    with _hbm_op("groupnorm_stats+apply", 6 * elems):                   # two reads + one write
        L.check(_lib.vx_groupnorm(_ptr(x1), c1, _ptr(x2), c2, frames, hw, groups, float(eps), _ptr(gamma), _ptr(beta),
                                  int(silu), _ptr(out), _ptr(ws), slices, width, pad, _stream()), "vx_groupnorm")
    return out


# (like LN_FOLD: the fold needs the persistent kernel, which needs the FAST addressing the VX_GEMM_NOFAST knob turns off)
GN_FOLD = [os.environ.get("VX_GN_FOLD", "1") != "0" and os.environ.get("VX_GEMM_NOFAST") is None]


def gn_fold_applies(m, hw, c, n):
    """Whether `groupnorm_fold_linear` + a grouped-weight GEMM replace groupnorm + GEMM for a [m, c] -> [m, n] layer:
    the switch is on, the persistent 256 x 320 kernel takes the launch (decided from batch-independent facts, like
    `_ring_hint`), a 256-row tile never straddles a frame, and the per-frame weight copies (frames x n x c) are cheaper
    to write than the normalised tensor (2 x m x c): c = 320 in practice (the 64x64 and 96x96 levels)."""
    items = _ITEMS[0]
    # (every deciding fact is in the key: the rows of ONE item included - ADVICE r05)
    geom = (("c", c), ("n", n), ("hw", hw), ("rows_item", m // items if items and items > 0 and m % items == 0 else None))
    folded, applied = "GroupNorm folded into per-frame weights (no normalised tensor)", "GroupNorm apply pass + GEMM"
    if not GN_FOLD[0] or items is None or items <= 0 or m % items or hw % 256 or n % 320 or c % 64:
        return _note_path("groupnorm_proj_in", geom, False, applied, "fold off or geometry not whole 256 x 320 tiles per frame")
    if _ring_off(c):
        # the persistent kernel is switched off (A/B knob ops.RING_MODE)
        return _note_path("groupnorm_proj_in", geom, False, applied, "persistent kernel switched off")
    rows_item = m // items
    ok = rows_item % 256 == 0 and (2 * rows_item // 256) * (n // 320) >= 192 and n * 2 <= hw
    return _note_path("groupnorm_proj_in", geom, ok, folded if ok else applied,
                      "" if ok else "per-frame weight copies would cost more than the normalised tensor")


def groupnorm_stats(x1, *, frames, hw, groups, x2=None):
    """The statistics of `groupnorm` alone -> (workspace, partial sums per frame) for `groupnorm_fold_linear`: the
    producer's own (`gn_of(x1)`) when it left them, else the statistics pass of `groupnorm`."""
    _chk_bf16(x1, "x1")
    st = gn_of(x1) if x2 is None else None
    if st is not None and st.fits(frames, hw, groups, x1.shape[-1]):
        return st.ws, st.slabs
    if not x1.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise ValueError("groupnorm inputs must be contiguous")
    c1 = x1.shape[-1]
    c2 = x2.shape[-1] if x2 is not None else 0
    slices = _gn_slices(hw)
    ws = torch.empty(int(_lib.vx_groupnorm_ws_floats(frames, slices, groups)), device=x1.device, dtype=torch.float32)
    with _hbm_op("groupnorm_stats", 2 * frames * hw * (c1 + c2)):
        L.check(_lib.vx_groupnorm_stats(_ptr(x1), c1, _ptr(x2), c2, frames, hw, groups, _ptr(ws), slices, _stream()),
                "vx_groupnorm_stats")
    return ws, slices


def groupnorm_fold_linear(ws, gamma, w, bias_beta, *, frames, hw, groups, eps, slices=None):
    """GroupNorm (no activation) folded into the linear layer behind it: -> (w_f bf16 [frames, N, C], b_f float32
    [frames, N]) with GN(x) w^T + b == x w_f[f]^T + b_f[f] on the pixels of frame f; bias_beta = b + w beta (float32,
    weights.fold_groupnorm).  Use: gemm(x, w_f, None, rowbias=b_f, rows_per_group=hw, w_group_rows=hw)."""
    _chk_bf16(w, "w")
    n, c = w.shape
    if bias_beta.dtype != torch.float32 or gamma.dtype != torch.float32 or not w.is_contiguous():
        raise TypeError("groupnorm_fold_linear: float32 gamma / bias_beta, contiguous bf16 weight expected")
    w_f = torch.empty((frames, n, c), device=w.device, dtype=L.ELEM[0])
    b_f = torch.empty((frames, n), device=w.device, dtype=torch.float32)
    L.check(_lib.vx_groupnorm_fold_linear(_ptr(ws), frames, hw, slices or _gn_slices(hw), groups, float(eps), _ptr(gamma), c,
                                          _ptr(w), _ptr(bias_beta), n, _ptr(w_f), _ptr(b_f), _stream()),
            "vx_groupnorm_fold_linear")
    return w_f, b_f


def layernorm(x, gamma, beta, eps=1e-5, *, add=None, add_rows_per_entry=1, add_entries=1, out=None):
    _chk_bf16(x, "x")
    ldx, rows = _row_stride(x)
    c = x.shape[-1]
    if out is None:
        out = torch.empty((rows, c), device=x.device, dtype=L.ELEM[0])
    with _hbm_op("layernorm", 4 * rows * c):
        L.check(_lib.vx_layernorm(_ptr(x), ldx, rows, c, float(eps), _ptr(gamma), _ptr(beta), _ptr(add),
                                  add_rows_per_entry, add_entries, _ptr(out), _row_stride(out)[0], _stream()),
                "vx_layernorm")
    return out


_BOUNDED_SOFTMAX = [os.environ.get("VX_ATTN_BOUND", "1") != "0"]


def key_norm_max(k, *, kv_batches, heads, n_kv, head_dim):
    """float32 [kv_batches * heads]: max over the keys of |k_j| per (kv batch, head) - the Kmax of the bounded softmax."""
    ldk, _ = _row_stride(k)
    out = torch.empty((kv_batches * heads,), device=k.device, dtype=torch.float32)
    with _hbm_op("key_norm_max", 2 * kv_batches * n_kv * heads * head_dim):
        L.check(_lib.vx_key_norm_max(_ptr(k), ldk, kv_batches, heads, n_kv, head_dim, _ptr(out), _stream()),
                "vx_key_norm_max")
    return out


def attention(q, k, vt, *, batch, heads, n_q, n_kv, head_dim, q_per_kv=1, out=None, kmax=None, k_prescaled=False):
    """q: [batch*n_q, *] view; k: [kv_batches*n_kv, *] view; vt: [kv_batches, heads, head_dim, pitch].
    k_prescaled: k already carries head_dim^-1/2 * log2(e) (weights.key_fold): the kernels take q . k as the base-2
    logit (vx_attention scale = 0).
    The 64x64 level's head dim (32 < d <= 48, not a multiple of 16: SD-1.5's d = 40) runs the bounded-softmax kernel
    (vx_attention_bounded; `kmax` = a precomputed key_norm_max of k, e.g. of a step-invariant reference bank)."""
    ldq, _ = _row_stride(q)
    ldk, _ = _row_stride(k)
    if out is None:
        out = torch.empty((batch * n_q, heads * head_dim), device=q.device, dtype=L.ELEM[0])
    scale = 0.0 if k_prescaled else head_dim ** -0.5
    # algorithmic work: QK^T + PV = 4 n_q n_kv d FLOP per (batch, head) at the TRUE head dim (zero padding is not counted);
    # q and the output once, K and V^T once per kv batch
    c = heads * head_dim
    flops = 4.0 * batch * heads * n_q * n_kv * head_dim
    nbytes = 2 * (2 * batch * n_q * c + 2 * (batch // q_per_kv) * n_kv * c)
    if _BOUNDED_SOFTMAX[0] and 32 < head_dim <= 48 and head_dim % 16:
        if kmax is None:
            kmax = key_norm_max(k, kv_batches=batch // q_per_kv, heads=heads, n_kv=n_kv, head_dim=head_dim)
        with _hbm_op("attention", nbytes, flops=flops):
            L.check(_lib.vx_attention_bounded(_ptr(q), ldq, _ptr(k), ldk, _ptr(vt), vt.shape[-1], _ptr(out),
                                              _row_stride(out)[0], batch, heads, n_q, n_kv, head_dim, q_per_kv,
                                              scale, _ptr(kmax), _stream()), "vx_attention_bounded")
        return out
    with _hbm_op("attention", nbytes, flops=flops):
        L.check(_lib.vx_attention(_ptr(q), ldq, _ptr(k), ldk, _ptr(vt), vt.shape[-1], _ptr(out), _row_stride(out)[0],
                                  batch, heads, n_q, n_kv, head_dim, q_per_kv, scale, _stream()),
                "vx_attention")
    return out


def temporal_attention(qkv, *, b, f, hw, heads, head_dim, out=None):
    """qkv: [(b f) hw, 3C] (Q | K | V columns) -> [(b f) hw, C]; attention runs over f per (b, pixel, head)."""
    _chk_bf16(qkv, "qkv")
    ld, rows = _row_stride(qkv)
    if rows != b * f * hw:
        raise ValueError("qkv rows != b*f*hw")
    if out is None:
        out = torch.empty((rows, heads * head_dim), device=qkv.device, dtype=L.ELEM[0])
    # reads q | k | v, writes out (bf16); 4 f f d FLOP per (batch, pixel, head)
    with _hbm_op("temporal_attention", 2 * rows * 4 * heads * head_dim, flops=4.0 * rows * f * heads * head_dim):
        L.check(_lib.vx_temporal_attention(_ptr(qkv), ld, _ptr(out), _row_stride(out)[0], b, f, hw, heads, head_dim,
                                           head_dim ** -0.5, _stream()), "vx_temporal_attention")
    return out


def small_kv_attention(q, kv, *, batch, n_q, n_kv, heads, head_dim, out=None):
    """q: [batch*n_q, C]; kv: [batch*n_kv, 2C] (K | V columns) -> [batch*n_q, C]."""
    ldq, _ = _row_stride(q)
    ldkv, _ = _row_stride(kv)
    c = heads * head_dim
    if out is None:
        out = torch.empty((batch * n_q, c), device=q.device, dtype=L.ELEM[0])
    with _hbm_op("small_kv_attention", 2 * (2 * batch * n_q * c + 2 * batch * n_kv * c)):   # q + out, K | V
        L.check(_lib.vx_small_kv_attention(_ptr(q), ldq, _ptr(kv), ldkv, c, _ptr(out), _row_stride(out)[0], batch, n_q,
                                           n_kv, heads, head_dim, head_dim ** -0.5, _stream()),
                "vx_small_kv_attention")
    return out


# Nearest-2x upsampling + conv3x3 as four 2x2 convolutions over the original image (weights.fold_upsample_phases, round 6):
# 2.25x fewer FLOPs than the convolution over the upsampled image.  VX_UPSAMPLE_PHASES=0 restores the one launch with the
# upsampling fused into its gather (A/B knob); VX_UPSAMPLE_PHASES_MIN_HW: smallest INPUT frame (pixels) that takes the four
# launches (below, the 2x2 launches are latency-bound split-K launches and the copies cost more than the FLOPs save).
UPSAMPLE_PHASES = [os.environ.get("VX_UPSAMPLE_PHASES", "1") != "0"]
UPSAMPLE_PHASES_MIN_HW = [int(os.environ.get("VX_UPSAMPLE_PHASES_MIN_HW", "256"))]


def upsample_phases_applies(H, W, c):
    geom = (("hw_in", H * W), ("c", c))
    four, one = "four 2x2 convolutions over the original image + interleave", "one 3x3 convolution with the upsampling in its gather"
    ok = UPSAMPLE_PHASES[0] and H * W >= UPSAMPLE_PHASES_MIN_HW[0] and c % 64 == 0
    return _note_path("upsample_conv", geom, ok, four if ok else one,
                      "" if ok else "switched off / input frame below VX_UPSAMPLE_PHASES_MIN_HW pixels / channels not a multiple of 64")


def pad_image(x, frames, H, W):
    """x [frames, H*W, C] -> the persistent zero-bordered image [frames, (H+2)*(W+2), C] of that shape (`padded_buffer`)."""
    _chk_bf16(x, "x")
    if not x.is_contiguous():
        raise ValueError("pad_image: contiguous input")
    c = x.shape[-1]
    out = padded_buffer(x.device, frames, H, W, c)
    L.check(_lib.vx_pad_image(_ptr(x), frames, H, W, c, _ptr(out), _stream()), "vx_pad_image")
    return out


def pixel_shuffle2x(phases, frames, H, W):
    """phases [4, frames*H*W, C] (phase a * 2 + b) -> [frames, 2H*2W, C] with out[f, 2y + a, 2x + b] = phases[a * 2 + b][f, y, x]."""
    _chk_bf16(phases, "phases")
    if phases.dim() != 3 or phases.shape[0] != 4 or phases.shape[1] != frames * H * W or not phases.is_contiguous():
        raise ValueError("pixel_shuffle2x: phases must be a contiguous [4, frames*H*W, C] tensor")
    c = phases.shape[-1]
    out = torch.empty((frames, 4 * H * W, c), device=phases.device, dtype=L.ELEM[0])
    L.check(_lib.vx_pixel_shuffle2x(_ptr(phases), phases.stride(0), frames, H, W, c, _ptr(out), _stream()), "vx_pixel_shuffle2x")
    return out


def upsample_conv_phases(x, w_phases, bias, *, frames, H, W):
    """Upsample (nearest x2) + conv3x3 of x [frames, H*W, Cin] -> [frames, 2H*2W, Cout] through the four phase weights of
    weights.fold_upsample_phases: one copy into the zero-bordered image, four pad-0 2x2 convolutions whose window starts
    (a, b) pixels into it, one interleave."""
    cout = w_phases.shape[1]
    xp = pad_image(x, frames, H, W)
    ph = torch.empty((4, frames * H * W, cout), device=x.device, dtype=L.ELEM[0])
    g = ConvGeom(frames, H + 2, W + 2, 2, 2, 1, 0, out_hw=(H, W))
    for a in (0, 1):
        for b in (0, 1):
            gemm(xp.view(frames * (H + 2) * (W + 2), -1), w_phases[a * 2 + b], bias, geom=g,
                 a_pixel_offset=a * (W + 2) + b, out=ph[a * 2 + b])
    return pixel_shuffle2x(ph, frames, H, W)


# proj_out folded into the feed-forward's second linear (weights.fold_ff_proj, round 6): out = [h | g] [Wp | Wp W2]^T + b +
# x_in as ONE dual-source GEMM of K = 5C.  VX_FF_PROJ_FOLD=0 restores the two launches (A/B knob); VX_FF_PROJ_FOLD_MIN_HW:
# smallest frame (rows) that takes the fold - at the 8x8 level the K = 6400 launch is a split-K launch, which cannot leave
# the next GroupNorm's partial sums.
FF_PROJ_FOLD = [os.environ.get("VX_FF_PROJ_FOLD", "1") != "0"]
FF_PROJ_FOLD_MIN_HW = [int(os.environ.get("VX_FF_PROJ_FOLD_MIN_HW", "256"))]
FF_PROJ_FOLD_MM = [os.environ.get("VX_FF_PROJ_FOLD_MM", "1") != "0"]      # the motion modules' blocks as well (A/B knob)


def ff_proj_fold_applies(m, c, hidden):
    """Batch-independent: the rows of one frame (ops.frame_rows), the widths and the switch."""
    hw = _FRAME_ROWS[0]
    geom = (("c", c), ("hidden", hidden), ("hw", hw))
    one, two = "FF output GEMM + proj_out as one dual-source GEMM (K = 5C)", "two launches (FF output GEMM, proj_out)"
    if not FF_PROJ_FOLD[0] or FP8_PROJ[0]:
        return _note_path("ff_proj_out", geom, False, two, "switched off / fp8 projections")
    ok = hw is not None and hw >= FF_PROJ_FOLD_MIN_HW[0] and c % 64 == 0 and hidden % 64 == 0
    return _note_path("ff_proj_out", geom, ok, one if ok else two, "" if ok else "frame below VX_FF_PROJ_FOLD_MIN_HW rows")


# The audio cross-attention of a spatial transformer block as ONE streaming launch (vx_audio_xattn, round 6): five audio
# tokens per frame -> q-projection, 5-key attention and out-projection collapse into two 48-column products with per-frame
# operands built once per clip (vx_audio_xattn_pack).  VX_AX_FUSED=0 restores the three launches (A/B knob).
AX_FUSED = [os.environ.get("VX_AX_FUSED", "1") != "0"]


class AudioFold:
    """Per-frame operands of `audio_xattn` for `frames` frames of width c (vx_audio_xattn_pack outputs)."""

    def __init__(self, kq, colsum, sbias, vo, frames, c):
        self.kq, self.colsum, self.sbias, self.vo, self.frames, self.c = kq, colsum, sbias, vo, frames, c


def audio_xattn_applies(c, heads, n_ctx, hw):
    geom = (("c", c), ("heads", heads), ("n_ctx", n_ctx), ("hw%16", hw % 16))
    one, three = "one launch (vx_audio_xattn)", "three launches (q GEMM + 5-key attention + output GEMM)"
    if not AX_FUSED[0] or not LN_FOLD[0] or FP8_PROJ[0]:
        return _note_path("audio_cross_attention", geom, False, three, "switched off / LayerNorm fold off / fp8 projections")
    ok = bool(_lib.vx_audio_xattn_supported(int(c), int(heads), int(n_ctx), int(hw)))
    return _note_path("audio_cross_attention", geom, ok, one if ok else three,
                      "" if ok else "the one-launch form is built for 8 heads, 5 audio tokens per frame, c % 320 == 0")


def audio_xattn_pack(kv, wq_folded, bq_folded, wo, *, frames, n_ctx, heads):
    """kv: [frames*n_ctx, 2C] (K | V: the to_k | to_v GEMM of the audio tokens), wq_folded / bq_folded: the LayerNorm-folded
    to_q weight [C, C] / bias [C] (weights.fold_layernorm), wo: to_out weight [C, C] -> AudioFold."""
    _chk_bf16(kv, "kv")
    _chk_bf16(wq_folded, "wq")
    _chk_bf16(wo, "wo")
    c = wo.shape[0]
    if kv.shape != (frames * n_ctx, 2 * c) or tuple(wq_folded.shape) != (c, c) or tuple(wo.shape) != (c, c) or \
            not wq_folded.is_contiguous() or not wo.is_contiguous():
        raise ValueError("audio_xattn_pack: shapes")
    if bq_folded is not None and (bq_folded.dtype != torch.float32 or bq_folded.numel() != c):
        raise TypeError("audio_xattn_pack: folded bias must be float32 [C]")
    dev = kv.device
    kq = torch.empty((frames, 48 * c), device=dev, dtype=L.ELEM[0])
    vo = torch.empty((frames, 48 * c), device=dev, dtype=L.ELEM[0])
    cs = torch.empty((frames, 48), device=dev, dtype=torch.float32)
    sb = torch.empty((frames, 48), device=dev, dtype=torch.float32)
    L.check(_lib.vx_audio_xattn_pack(_ptr(kv), _row_stride(kv)[0], _ptr(wq_folded), _ptr(bq_folded), _ptr(wo), c, heads, n_ctx,
                                     frames, _ptr(kq), _ptr(cs), _ptr(sb), _ptr(vo), _stream()), "vx_audio_xattn_pack")
    return AudioFold(kq, cs, sb, vo, frames, c)


def audio_xattn(h, stats, fold, bias_o, alpha, *, rows_per_frame, stats_out=None, stats_eps=1e-5, out=None):
    """h [frames*rows_per_frame, C] <- h + alpha * attn2(LayerNorm(h), audio tokens) in one launch (in place unless `out`).
    stats: the LayerNorm statistics of h's rows ([m, 2] or [m, 4], as ops.gemm(ln=...)); stats_out: receives the statistics of
    the rows written (same formats; may be `stats`)."""
    _chk_bf16(h, "h")
    ldx, m = _row_stride(h)
    c = h.shape[-1]
    if out is None:
        out = h
    if m != fold.frames * rows_per_frame or c != fold.c:
        raise ValueError(f"audio_xattn: {m} rows of width {c} against operands for {fold.frames} frames x {rows_per_frame} rows, width {fold.c}")
    if stats.dtype != torch.float32 or not stats.is_contiguous() or tuple(stats.shape) not in ((m, 2), (m, 4)):
        raise ValueError("audio_xattn: stats must be a contiguous float32 [m, 2] / [m, 4] tensor")
    p = L.AxAttnParams()
    p.x, p.ldx, p.out, p.ldo = h.data_ptr(), ldx, out.data_ptr(), _row_stride(out)[0]
    p.rows, p.c, p.rows_per_frame = m, c, rows_per_frame
    p.ln_stats, p.ln_stats_parts, p.ln_eps = stats.data_ptr(), 2 if stats.shape[1] == 4 else 0, 1e-5
    p.kq, p.kq_colsum, p.kq_bias, p.vo = fold.kq.data_ptr(), fold.colsum.data_ptr(), fold.sbias.data_ptr(), fold.vo.data_ptr()
    p.bias_o, p.alpha = bias_o.data_ptr(), float(alpha)
    if stats_out is not None:
        if stats_out.dtype != torch.float32 or not stats_out.is_contiguous() or tuple(stats_out.shape) not in ((m, 2), (m, 4)):
            raise ValueError("audio_xattn: stats_out must be a contiguous float32 [m, 2] / [m, 4] tensor")
        p.row_stats_out, p.row_stats_parts, p.row_stats_eps = stats_out.data_ptr(), 2 if stats_out.shape[1] == 4 else 0, float(stats_eps)
    # algorithmic bytes: the rows read once and written once, the per-frame operands once
    with _hbm_op("audio_xattn", 2 * (2 * m * c + 2 * fold.frames * 48 * c)):
        L.check(_lib.vx_audio_xattn(C.byref(p), _stream()), "vx_audio_xattn")
    _set_gn(out)
    return out


def add_row_bias(x, bias, alpha=1.0):
    ldx, rows = _row_stride(x)
    L.check(_lib.vx_add_row_bias(_ptr(x), ldx, rows, x.shape[-1], _ptr(bias), float(alpha), _stream()),
            "vx_add_row_bias")
    return x


def add_residual_f32(x, y32, out=None):
    """element(x + y32): x [rows, C] elements, y32 [rows, C] float32 (a GEMM's out_f32 rows) -> [rows, C] elements - the
    residual add and the one rounding of vx_gemm's STORE epilogue, for rows that travelled as float32 in between."""
    _chk_bf16(x, "x")
    if y32.dtype != torch.float32 or y32.shape != x.shape or y32.stride(-1) != 1:
        raise TypeError("add_residual_f32: y32 must be float32 of x's shape")
    ldx, rows = _row_stride(x)
    ldy, _ = _row_stride(y32)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=L.ELEM[0])
    L.check(_lib.vx_add_residual_f32(_ptr(x), ldx, _ptr(y32), ldy, rows, x.shape[-1], _ptr(out), _row_stride(out)[0], _stream()),
            "vx_add_residual_f32")
    _set_gn(out)
    return out


def gather_latents(latents, frame_ids, reps, c_pad=8):
    """latents fp32 [1, C, F, h, w], frame_ids int32 [f] (device) -> bf16 [reps*f, h*w, c_pad]."""
    _, c, F, h, w = latents.shape
    f = frame_ids.numel()
    out = torch.empty((reps * f, h * w, c_pad), device=latents.device, dtype=L.ELEM[0])
    L.check(_lib.vx_gather_latents(_ptr(latents), c, F, h * w, _ptr(frame_ids), f, reps, c_pad, _ptr(out),
                                   _stream()), "vx_gather_latents")
    return out


def cfg_combine(unet_out, c, f, hw, guidance, pred_slot):
    """unet_out fp32 [2f*hw, ld] -> pred_slot fp32 [c, f, hw] = u + s (c - u)."""
    L.check(_lib.vx_cfg_combine(_ptr(unet_out), unet_out.stride(0), c, f, hw, float(guidance), _ptr(pred_slot),
                                _stream()), "vx_cfg_combine")


def pack_rows(src, c, dst):
    """src fp32 [rows, ld] -> dst fp32 [rows, c] (dense): this rank's conv_out rows into its send slots."""
    if src.dtype != torch.float32 or dst.dtype != torch.float32 or not dst.is_contiguous():
        raise TypeError("pack_rows: float32 tensors, contiguous destination")
    rows = src.shape[0]
    if dst.numel() != rows * c:
        raise ValueError("pack_rows: destination size mismatch")
    L.check(_lib.vx_pack_rows(_ptr(src), src.stride(0), rows, c, _ptr(dst), _stream()), "vx_pack_rows")


def combine_units(gathered, unit_index, c, f, hw, guidance, preds):
    """gathered fp32 [units_total, (f/S)*hw, c]; unit_index int32 [nW, halves, S] -> preds fp32 [nW, c, f, hw]."""
    nW, halves, S = unit_index.shape
    if unit_index.dtype != torch.int32 or not unit_index.is_contiguous() or not gathered.is_contiguous():
        raise TypeError("combine_units: contiguous int32 index / contiguous gathered buffer expected")
    L.check(_lib.vx_combine_units(_ptr(gathered), _ptr(unit_index), nW, halves, S, c, f, hw, float(guidance),
                                  _ptr(preds), _stream()), "vx_combine_units")


def overlap_ddim_step(latents, preds, terms, frame_ids, counts, coef):
    """latents fp32 [1,C,F,h,w] updated in place for `frame_ids`; preds fp32 [slots, C, f, hw]."""
    _, c, F, h, w = latents.shape
    n = frame_ids.numel()
    L.check(_lib.vx_overlap_ddim_step(_ptr(latents), c, F, h * w, _ptr(preds), preds.shape[2], _ptr(terms),
                                      terms.shape[1], _ptr(frame_ids), _ptr(counts), n, *[float(v) for v in coef],
                                      _stream()), "vx_overlap_ddim_step")


def ncfhw_to_nhwc(x, c_pad=None):
    """fp32 [b, C, f, h, w] -> bf16 [(b f), h*w, c_pad]."""
    b, c, f, h, w = x.shape
    c_pad = c_pad or (c + 7) // 8 * 8
    x = x.contiguous().float()
    out = torch.empty((b * f, h * w, c_pad), device=x.device, dtype=L.ELEM[0])
    L.check(_lib.vx_ncfhw_to_nhwc(_ptr(x), b, c, f, h * w, c_pad, _ptr(out), _stream()), "vx_ncfhw_to_nhwc")
    return out


def nhwc_to_ncfhw(x, b, c, f, h, w):
    """fp32 [(b f)*hw, ld] -> fp32 [b, c, f, h, w]."""
    out = torch.empty((b, c, f, h, w), device=x.device, dtype=torch.float32)
    L.check(_lib.vx_nhwc_to_ncfhw(_ptr(x), x.stride(0), b, c, f, h * w, _ptr(out), _stream()), "vx_nhwc_to_ncfhw")
    return out


def median3d(video, want_f32=True, want_u8=False):
    """video fp32 [C, F, H, W] (device) -> (filtered fp32 [C, F, H, W] or None, uint8 [F, H, W, C] or None)."""
    if video.dtype != torch.float32 or not video.is_cuda or not video.is_contiguous():
        raise TypeError("median3d: expected a contiguous CUDA float32 [C, F, H, W] tensor")
    c, f, h, w = video.shape
    out = torch.empty_like(video) if want_f32 else None
    u8 = torch.empty((f, h, w, c), device=video.device, dtype=torch.uint8) if want_u8 else None
    L.check(_lib.vx_median3d(_ptr(video), c, f, h, w, _ptr(out), _ptr(u8), _stream()), "vx_median3d")
    return out, u8


def wave_conv1d(wave, wt, stride):
    """First wav2vec2 feature-encoder conv: wave float32 [samples], wt float32 [taps, C] -> bf16 [t_out, C]."""
    if wave.dtype != torch.float32 or wt.dtype != torch.float32 or not wave.is_contiguous() or not wt.is_contiguous():
        raise TypeError("wave_conv1d: contiguous float32 tensors expected")
    taps, c = wt.shape
    t_out = (wave.numel() - taps) // stride + 1
    if t_out < 1:
        raise ValueError(f"wave_conv1d: {wave.numel()} samples are shorter than one {taps}-tap window")
    out = torch.empty((t_out, c), device=wave.device, dtype=L.ELEM[0])
    L.check(_lib.vx_wave_conv1d(_ptr(wave), wave.numel(), _ptr(wt), c, taps, stride, _ptr(out), _stream()),
            "vx_wave_conv1d")
    return out


def vae_postprocess(x, n, c, h, w):
    """fp32 [n*hw, ld] -> fp32 [n, c, h, w] = clamp(x/2 + 0.5, 0, 1)."""
    out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    L.check(_lib.vx_vae_postprocess(_ptr(x), x.stride(0), n, c, h * w, _ptr(out), _stream()), "vx_vae_postprocess")
    return out
