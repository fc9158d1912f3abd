"""Host side of tools/conv3/libvx_conv3.so (round 5's one-pass GroupNorm + SiLU + 3x3 convolution; see vx_conv3.h): the ctypes
mirror of `vx_conv3_params` and the tensor-level wrappers that used to live in v_express_amd/ops.py while the kernel was part
of the shipped ABI.  Needs the tool library (tools/conv3/build_conv3_variants.sh; VX_CONV3_LIBRARY overrides the path)."""
import ctypes
import os

import torch

from v_express_amd import ops

C = ctypes
HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = [None]


class Conv3Params(C.Structure):
    """Mirror of `vx_conv3_params` (tools/conv3/vx_conv3.h)."""
    _fields_ = [
        ("x1", C.c_void_p), ("x2", C.c_void_p),
        ("c1", C.c_int32), ("c2", C.c_int32), ("ldx1", C.c_int32), ("ldx2", C.c_int32),
        ("frames", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("w_perm", C.c_void_p), ("n", C.c_int32),
        ("ab", C.c_void_p), ("ab_ld", C.c_int32), ("silu", C.c_int32),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p), ("rowbias_ld", C.c_int32), ("rows_per_group", C.c_int32),
        ("residual", C.c_void_p), ("ldr", C.c_int32),
        ("out", C.c_void_p), ("ldc", C.c_int32),
        ("gn_ws", C.c_void_p), ("gn_groups", C.c_int32), ("gn_hw", C.c_int32),
    ]



def _c3():
    if _LIB[0] is None:
        path = os.environ.get("VX_CONV3_LIBRARY", os.path.join(HERE, "libvx_conv3.so"))
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run tools/conv3/build_conv3_variants.sh")
        lib = C.CDLL(path)
        i32, f32, vp = C.c_int32, C.c_float, C.c_void_p
        lib.vx_conv3x3_gn.argtypes = [C.POINTER(Conv3Params), vp]
        lib.vx_conv3x3_gn_supported.argtypes = [C.POINTER(Conv3Params)]
        lib.vx_groupnorm_scale_shift.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp, i32, vp, i32, vp]
        for fn in (lib.vx_conv3x3_gn, lib.vx_conv3x3_gn_supported, lib.vx_groupnorm_scale_shift):
            fn.restype = i32
        _LIB[0] = lib
    return _LIB[0]


# GroupNorm + SiLU + 3x3 convolution of a resnet block as ONE pass over the raw tensor (csrc/vx_conv3.hip, round 5): the
# normalised, zero-bordered copy that groupnorm(pad_hw=...) + gemm(3x3) went through is never written, and a tile's
# activations cross the CU's L1 once per 32-channel chunk instead of nine times.  Correct (kernel, model and full-size
# parity) but NOT faster than the two launches it replaces: 250 - 260 us against 220 + 34 us at the 64x64 level, whole path
# -0.8 ... -1.5 % in same-box A/B (profiles/r05b ... r05g: the in-LDS normalisation lengthens the L slots of four of every
# nine K-tiles by what the apply pass cost, and moved into the M slots its v_exp / v_rcp do not hide under the same
# wave's MFMAs).  OFF by default; VX_CONV3_GN=1 turns it on.
_C3_W = {}
C3_AB_LD = 1024


def conv3_weight(w):
    """[N, 9 C] conv weight with K = (ky, kx, c) -> the kernel's K order (32-channel chunk, tap, 32); cached per tensor."""
    key = (w.data_ptr(), tuple(w.shape))
    hit = _C3_W.get(key)
    if hit is None:
        n, k = w.shape
        c = k // 9
        wp = w.view(n, 9, c // 32, 32).permute(0, 2, 1, 3).reshape(n, k).contiguous()
        hit = _C3_W[key] = (w, wp)                     # keep `w` alive: the key is its address
    return hit[1]


def _conv3_params(x1, x2, frames, H, W, n):
    p = Conv3Params()
    p.x1, p.c1, p.ldx1 = x1.data_ptr(), x1.shape[-1], x1.stride(-2)
    if x2 is not None:
        p.x2, p.c2, p.ldx2 = x2.data_ptr(), x2.shape[-1], x2.stride(-2)
    p.frames, p.h, p.w, p.n, p.ab_ld = frames, H, W, n, C3_AB_LD
    return p


def conv3_gn_supported(H, W, c_in, n, c1=None):
    """Geometries vx_conv3x3_gn takes (vx_conv3x3_gn_supported): W = 64 or 32 with whole 256-pixel tiles per frame, source
    channel counts multiples of 32 summing to a multiple of 64 (<= 1024), output channels a multiple of 320."""
    c1 = c_in if c1 is None else c1
    return (W in (64, 32) and (H * W) % 256 == 0 and c1 % 32 == 0 and (c_in - c1) % 32 == 0 and c_in % 64 == 0
            and c_in <= C3_AB_LD and n % 320 == 0)


def groupnorm_scale_shift(ws, slices, gamma, beta, *, frames, hw, groups, eps):
    """float32 [frames, 1024, 2] = (scale, shift) per frame and channel from GroupNorm partial sums (vx_groupnorm_scale_shift)."""
    c = gamma.numel()
    ab = torch.empty((frames, C3_AB_LD, 2), device=ws.device, dtype=torch.float32)
    ops.L.check(_c3().vx_groupnorm_scale_shift(ops._ptr(ws), int(slices), frames, hw, groups, float(eps), ops._ptr(gamma), ops._ptr(beta), c,
                                          ops._ptr(ab), C3_AB_LD, ops._stream()), "vx_groupnorm_scale_shift")
    return ab


def conv3_gn(x1, gamma, beta, w, bias, *, frames, H, W, groups, eps, x2=None, silu=True, rowbias=None, rows_per_group=0,
             residual=None, gn=None):
    """out = residual + conv3x3_pad1(act(GroupNorm(x1 | x2))) + bias + rowbias, act = SiLU (silu=True): ResnetBlock3D's
    norm -> SiLU -> conv (modules/resnet.py:220-223, :235-244) without the normalised intermediate.
    x1: [frames, H*W, C1] (+ x2: [frames, H*W, C2], the skip concat); w: [N, 9 (C1 + C2)] with K = (ky, kx, c).
    The statistics come from the producer of x1 when it left them (`gn_of`), else from vx_groupnorm_stats.
    gn=(groups, hw): as in `gemm` - the epilogue leaves the next GroupNorm's partial sums on the returned tensor."""
    ops._chk_bf16(x1, "x1")
    if not x1.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise ValueError("conv3_gn inputs must be contiguous")
    hw = H * W
    n = w.shape[0]
    ws, slices = ops.groupnorm_stats(x1, frames=frames, hw=hw, groups=groups, x2=x2)
    ab = groupnorm_scale_shift(ws, slices, gamma, beta, frames=frames, hw=hw, groups=groups, eps=eps)
    p = _conv3_params(x1, x2, frames, H, W, n)
    wp = conv3_weight(w)
    p.w_perm, p.ab, p.silu = wp.data_ptr(), ab.data_ptr(), int(bool(silu))
    if bias is not None:
        if bias.dtype != torch.float32:
            raise TypeError("bias must be float32")
        p.bias = bias.data_ptr()
    if rowbias is not None:
        if rowbias.dtype != torch.float32 or rowbias.stride(-1) != 1:
            raise TypeError("rowbias must be float32 with contiguous columns")
        p.rowbias, p.rowbias_ld, p.rows_per_group = rowbias.data_ptr(), rowbias.stride(0), rows_per_group
    out = torch.empty((frames * hw, n), device=x1.device, dtype=ops.BF16)
    p.out, p.ldc = out.data_ptr(), n
    if residual is not None:
        ops._chk_bf16(residual, "residual")
        p.residual, p.ldr = residual.data_ptr(), ops._row_stride(residual)[0]
    gst = None
    if gn is not None and ops.GN_FUSED[0]:
        g_groups, g_hw = gn
        cg = n // g_groups if g_groups else 0
        if g_groups > 0 and n % g_groups == 0 and (frames * hw) % g_hw == 0 and g_hw % 128 == 0 and cg and 80 % cg == 0:
            slabs = g_hw // 128
            gst = ops.GnStats(torch.empty((frames * hw // g_hw, slabs, g_groups, 2), device=x1.device, dtype=torch.float32),
                          slabs, g_groups, frames * hw // g_hw, g_hw, n)
            p.gn_ws, p.gn_groups, p.gn_hw = gst.ws.data_ptr(), int(g_groups), int(g_hw)
    cin = p.c1 + p.c2
    # reads the raw rows once, the weights once, writes the output once (+ residual); 2 m n 9 cin FLOP
    with ops._hbm_op("conv3_gn", 2 * (frames * hw * (cin + n * (2 if residual is not None else 1)) + n * 9 * cin),
                 flops=2.0 * frames * hw * n * 9 * cin):
        ops.L.check(_c3().vx_conv3x3_gn(ctypes.byref(p), ops._stream()), "vx_conv3x3_gn")
    ops._set_gn(out, gst)
    return out


