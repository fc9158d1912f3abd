"""ctypes binding of libvexpress_hip.so (C ABI: include/vexpress_hip.h).

There is NO fallback: if the library is missing, unbuildable or lacks a symbol, importing this module
raises.  The library is built in-tree (v-express_amd/libvexpress_hip.so) by `__graft_entry__.build()` /
`make -C v-express_amd/csrc`, so it travels with the repo snapshot to the GPU box.
"""
import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# VX_LIBRARY: another build of the SAME library (same-box A/B runs of kernel variants, tools/ only); default in-tree
LIB_PATH = os.environ.get("VX_LIBRARY") or os.path.join(_HERE, "libvexpress_hip.so")
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "vexpress_hip.h")

VX_EPI_STORE, VX_EPI_GEGLU, VX_EPI_SPLIT = 0, 1, 2
VX_PART_ROWS, VX_PART_VT = 0, 1
VX_ACT_NONE, VX_ACT_SILU, VX_ACT_GELU = 0, 1, 2


class FfParams(C.Structure):
    """Mirror of `vx_ff_params` (include/vexpress_hip.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int32), ("m", C.c_int32), ("c", C.c_int32), ("hidden", C.c_int32),
        ("w1t", C.c_void_p), ("w2t", C.c_void_p), ("bias1", C.c_void_p), ("ln_colsum", C.c_void_p),
        ("ln_stats", C.c_void_p), ("bias2", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_int32),
        ("out", C.c_void_p), ("ldo", C.c_int32),
    ]


class AxAttnParams(C.Structure):
    """Mirror of `vx_axattn_params` (include/vexpress_hip.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int32), ("out", C.c_void_p), ("ldo", C.c_int32), ("rows", C.c_int32), ("c", C.c_int32),
        ("rows_per_frame", C.c_int32), ("ln_stats", C.c_void_p), ("ln_stats_parts", C.c_int32), ("ln_eps", C.c_float),
        ("kq", C.c_void_p), ("kq_colsum", C.c_void_p), ("kq_bias", C.c_void_p), ("vo", C.c_void_p), ("bias_o", C.c_void_p),
        ("alpha", C.c_float), ("row_stats_out", C.c_void_p), ("row_stats_parts", C.c_int32), ("row_stats_eps", C.c_float),
    ]


class TBlockParams(C.Structure):
    """Mirror of `vx_tblock_params` (include/vexpress_hip.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int32), ("b", C.c_int32), ("f", C.c_int32), ("hw", C.c_int32), ("c", C.c_int32),
        ("heads", C.c_int32), ("wqkv_t", C.c_void_p), ("wo_t", C.c_void_p), ("colsum_p", C.c_void_p), ("bias_o", C.c_void_p), ("ln_stats", C.c_void_p), ("stats_out", C.c_void_p), ("ln_eps", C.c_float),
        ("scale", C.c_float),
    ]


class GemmParams(C.Structure):
    """Mirror of `vx_gemm_params` (include/vexpress_hip.h).""" 
    _fields_ = [
        ("a", C.c_void_p), ("a2", C.c_void_p),
        ("c1", C.c_int32), ("c2", C.c_int32),
        ("lda1", C.c_int32), ("lda2", C.c_int32),
        ("nb", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("upsample", C.c_int32),
        ("h_out", C.c_int32), ("w_out", C.c_int32),
        ("w", C.c_void_p),
        ("n", C.c_int32), ("k", C.c_int32), ("m", C.c_int32),
        ("epi", C.c_int32), ("act", C.c_int32),
        ("alpha", C.c_float),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p),
        ("rowbias_ld", C.c_int32), ("rows_per_group", C.c_int32),
        ("residual", C.c_void_p), ("ldr", C.c_int32),
        ("out", C.c_void_p), ("ldc", C.c_int32), ("out_f32", C.c_int32),
        ("part_cols", C.c_int32), ("n_parts", C.c_int32),
        ("part_out", C.c_void_p * 3), ("part_kind", C.c_int32 * 3), ("part_ld", C.c_int32 * 3),
        ("seq_len", C.c_int32), ("head_dim", C.c_int32), ("vt_pitch", C.c_int32),
        ("splitk", C.c_int32), ("splitk_ws", C.c_void_p),
        ("ring_hint", C.c_int32),
        ("a_fp8", C.c_int32), ("a_scale", C.c_void_p), ("w_scale", C.c_void_p),
        ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p),
        ("row_stats_out", C.c_void_p), ("row_stats_eps", C.c_float),
        ("w_group_rows", C.c_int32),
        ("gn_ws", C.c_void_p), ("gn_groups", C.c_int32), ("gn_hw", C.c_int32),
        ("row_stats_parts", C.c_int32), ("ln_stats_parts", C.c_int32), ("ln_eps", C.c_float),
        ("coop_epoch", C.c_int32),
    ]


def declared_symbols():
    """Every function name declared in include/vexpress_hip.h."""
    with open(HEADER) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vx_[a-z0-9_]+)\s*\(", text)))


def build(force=False):
    """Compile the HIP sources for gfx950 (cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    r = subprocess.run(["make", "-C", CSRC, "-j", str(os.cpu_count() or 4)], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError("building libvexpress_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def _load(path=None, element="bf16"):
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `make -C v-express_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{path} does not export {missing}; rebuild it")
    i32, f32, vp, i64 = C.c_int32, C.c_float, C.c_void_p, C.c_int64
    lib.vx_last_error_string.restype = C.c_char_p
    lib.vx_abi_version.restype = i32
    lib.vx_device_info.argtypes = [i32, C.POINTER(i32)]
    lib.vx_gemm.argtypes = [C.POINTER(GemmParams), vp]
    lib.vx_gemm_config_name.argtypes = [C.POINTER(GemmParams)]
    lib.vx_gemm_config_name.restype = C.c_char_p
    lib.vx_gemm_last_kernel.restype = C.c_char_p
    lib.vx_last_kernel.restype = C.c_char_p
    lib.vx_build_id.restype = C.c_char_p
    lib.vx_element_type.restype = C.c_char_p
    lib.vx_gemm_splitk_ws_bytes.argtypes = [i32, i32, i32]
    lib.vx_gemm_splitk_ws_bytes.restype = i64
    lib.vx_gemm_ring_coop_ok.argtypes = [C.POINTER(GemmParams)]
    lib.vx_groupnorm_ws_floats.restype = i64
    lib.vx_groupnorm_ws_floats.argtypes = [i32, i32, i32]
    lib.vx_groupnorm.argtypes = [vp, i32, vp, i32, i32, i32, i32, f32, vp, vp, i32, vp, vp, i32, i32, i32, vp]
    lib.vx_groupnorm_stats.argtypes = [vp, i32, vp, i32, i32, i32, i32, vp, i32, vp]
    lib.vx_groupnorm_apply.argtypes = [vp, i32, vp, i32, i32, i32, i32, f32, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp]
    lib.vx_gemm_gn_slabs.argtypes = [C.POINTER(GemmParams)]
    lib.vx_ff_fused.argtypes = [C.POINTER(FfParams), vp]
    lib.vx_ff_pack_weights.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    lib.vx_tblock_fused.argtypes = [C.POINTER(TBlockParams), vp]
    lib.vx_tblock_pack.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.vx_tblock_packed_bytes.argtypes = [i32]
    lib.vx_tblock_packed_bytes.restype = i64
    lib.vx_audio_xattn_packed_bytes.argtypes = [i32, i32]
    lib.vx_audio_xattn_packed_bytes.restype = i64
    lib.vx_audio_xattn_supported.argtypes = [i32, i32, i32, i32]
    lib.vx_audio_xattn_pack.argtypes = [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.vx_audio_xattn.argtypes = [C.POINTER(AxAttnParams), vp]
    lib.vx_groupnorm_fold_linear.argtypes = [vp, i32, i32, i32, i32, f32, vp, i32, vp, vp, i32, vp, vp, vp]
    lib.vx_layernorm.argtypes = [vp, i32, i32, i32, f32, vp, vp, vp, i32, i32, vp, i32, vp]
    lib.vx_row_stats.argtypes = [vp, i32, i32, i32, f32, vp, vp]
    lib.vx_row_stats_parts.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.vx_row_stats_finalize.argtypes = [vp, i32, i32, f32, vp, vp]
    lib.vx_layernorm_fp8.argtypes = [vp, i32, i32, i32, f32, vp, vp, vp, i32, i32, vp, i32, vp, vp]
    lib.vx_attention.argtypes = [vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp]
    lib.vx_attention_bounded.argtypes = [vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp]
    lib.vx_key_norm_max.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    lib.vx_temporal_attention.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, i32, f32, vp]
    lib.vx_small_kv_attention.argtypes = [vp, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, f32, vp]
    lib.vx_add_row_bias.argtypes = [vp, i32, i32, i32, vp, f32, vp]
    lib.vx_add_residual_f32.argtypes = [vp, i32, vp, i32, i32, i32, vp, i32, vp]
    lib.vx_pad_image.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.vx_pixel_shuffle2x.argtypes = [vp, i64, i32, i32, i32, i32, vp, vp]
    lib.vx_gather_latents.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, vp, vp]
    lib.vx_cfg_combine.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp]
    lib.vx_pack_rows.argtypes = [vp, i32, i64, i32, vp, vp]
    lib.vx_combine_units.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp]
    lib.vx_overlap_ddim_step.argtypes = [vp, i32, i32, i32, vp, i32, vp, i32, vp, vp, i32, f32, f32, f32, f32, vp]
    lib.vx_ncfhw_to_nhwc.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    lib.vx_nhwc_to_ncfhw.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    lib.vx_vae_postprocess.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.vx_median3d.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp]
    lib.vx_wave_conv1d.argtypes = [vp, i32, vp, i32, i32, i32, vp, vp]
    for name in declared_symbols():
        fn = getattr(lib, name)
        if name not in ("vx_last_error_string", "vx_groupnorm_ws_floats", "vx_gemm_config_name",
                        "vx_gemm_splitk_ws_bytes", "vx_gemm_last_kernel", "vx_last_kernel", "vx_build_id",
                        "vx_tblock_packed_bytes", "vx_element_type", "vx_audio_xattn_packed_bytes"):
            fn.restype = i32
    if lib.vx_abi_version() != 15:
        raise ImportError(f"{os.path.basename(path)} ABI version mismatch")
    if lib.vx_element_type().decode() != element:
        raise ImportError(f"{path} computes on {lib.vx_element_type().decode()} elements, expected {element}")
    return lib


lib = _load()

# ---- the element type of the running computation.  libvexpress_hip.so computes on bfloat16 storage, libvexpress_hip_f16.so
# (the same sources, -DVX_ELEM_F16, the same ABI) on IEEE half - the reference's default `--dtype fp16`
# (inference.py:44,150-151).  Which one a call goes to is decided by the MODEL's dtype: every model entry point runs inside
# `with element_type(model element)` (module_base.DeviceModule), ops.py allocates in ELEM[0] and calls `current()`.
import torch  # noqa: E402

LIB16_PATH = os.environ.get("VX_LIBRARY_F16") or os.path.join(_HERE, "libvexpress_hip_f16.so")
ELEM = [torch.bfloat16]
_LIB16 = [None]


def lib_f16():
    """The IEEE-half build, loaded on first use (a missing / stale / mismatching library is an ImportError, like the bf16 one)."""
    if _LIB16[0] is None:
        l16 = _load(LIB16_PATH, "f16")
        if not os.environ.get("VX_LIBRARY_F16") and not os.environ.get("VX_LIBRARY"):
            a, b = l16.vx_build_id().decode(), lib.vx_build_id().decode()
            if a != b:
                raise ImportError(f"{LIB16_PATH} ({a}) and {LIB_PATH} ({b}) were built from different sources: rebuild both")
        _LIB16[0] = l16
    return _LIB16[0]


def current():
    """The library of the element type in force (ELEM[0])."""
    return lib if ELEM[0] is torch.bfloat16 else lib_f16()


class element_type:
    """`with element_type(torch.float16):` - calls and allocations of ops.py inside the block use the IEEE-half library."""

    def __init__(self, dtype):
        if dtype not in (torch.bfloat16, torch.float16):
            raise TypeError(f"the kernels compute on bfloat16 or float16 elements, not {dtype}")
        self.dtype = dtype

    def __enter__(self):
        self.prev = ELEM[0]
        ELEM[0] = self.dtype
        if self.dtype is torch.float16:
            lib_f16()
        return self

    def __exit__(self, *a):
        ELEM[0] = self.prev


def source_id():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from - csrc/*.hip / *.h / *.cpp and the
    C ABI header, in name order.  (Of the sources, not of the .so: a rebuild on another machine need not be
    byte-identical, the sources it was built from are.  tools/lib_id.py prints the same value without importing torch.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                   glob.glob(os.path.join(CSRC, "*.cpp")) + [os.path.join(CSRC, "Makefile"), HEADER])
    try:
        for path in files:
            h.update(os.path.basename(path).encode())
            with open(path, "rb") as f:
                h.update(f.read())
    except OSError:
        return None                      # a binary-only install: nothing to compare the library with
    return h.hexdigest()[:16]


def _build_identity():
    """Identity of the LOADED binary: `vx_build_id()` = "<source hash the Makefile stamped>|<extra -D flags>".
    * stamped, no extra flags, hash == the sources on disk: that hash (what committed profiles are keyed by);
    * stamped with extra -D flags (make EXTRA_DEFS=...): "<hash>+<flags>" - never equal to a plain hash;
    * loaded through VX_LIBRARY (A/B builds of tools/build_*_variants.sh): "variant:<file name>"; unstamped: "unstamped:<file name>";
    * stamped with ANOTHER hash than the sources on disk: the .so is stale -> ImportError (VX_ALLOW_STALE_LIB=1: a warning
      and the identity "stale:<hash>", so that no committed measurement is paired with it)."""
    src, _, defs = lib.vx_build_id().decode().partition("|")
    if os.environ.get("VX_LIBRARY"):
        # an explicitly chosen A/B build (tools/build_*_variants.sh link the product's stamped vx_api.o): never the product
        return "variant:" + os.path.basename(LIB_PATH)
    if src == "unstamped" or len(src) != 16 or any(ch not in "0123456789abcdef" for ch in src):
        # (an empty or malformed stamp - tools/lib_id.py failed inside make - is "unstamped" too, never "stale": ADVICE r05)
        return "unstamped:" + os.path.basename(LIB_PATH)
    disk = source_id()
    if disk is not None and disk != src:
        msg = (f"{LIB_PATH} was built from kernel sources {src}, the sources on disk are {disk}: rebuild it "
               "(make -C v-express_amd/csrc, or __graft_entry__.build())")
        if os.environ.get("VX_ALLOW_STALE_LIB") != "1":
            raise ImportError(msg)
        import warnings
        warnings.warn(msg)
        return "stale:" + src
    return src + ("+" + defs if defs else "")


# identity of the kernel build: measurements committed under profiles/ carry it, and bench.py quotes a committed trace /
# counter file next to a live number only when it was taken with the SAME binary (`vx_build_id`, stamped by the Makefile)
LIB_SHA256 = _build_identity()


class VxError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = current().vx_last_error_string().decode("utf-8", "replace")
        raise VxError(f"libvexpress_hip {what} failed (rc={rc}): {msg}")
