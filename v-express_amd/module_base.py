"""torch.nn.Module-like surface shared by the v_express_amd model classes (they are not nn.Modules: weights live in
kernel-specific device layouts).  What the reference's call sites use is here: `.to(device / dtype)`, `.device`,
`.dtype`, `.eval()`, `.requires_grad_()`, `load_state_dict(sd, strict)`, `state_dict()`, `parameters()`
(inference.py:77-129,150-163; pipelines/v_express_pipeline.py:345).

Compute dtype = the model's dtype, as in the reference: `torch.bfloat16` runs libvexpress_hip.so (bf16 storage, fp32
accumulation, v_mfma_f32_16x16x32_bf16), `torch.float16` - the reference's default, `inference.py:44,150-151` - runs
libvexpress_hip_f16.so, the same kernel sources compiled for IEEE half (v_mfma_f32_16x16x32_f16: same MFMA rate, 11 instead
of 8 mantissa bits, 5 instead of 8 exponent bits).  `torch.float32` is an I/O dtype only (bf16 elements underneath).  Every
model entry point runs inside `lib.element_type(self._elem)`; the device layouts of the weights are (re)built in that
element type, from the source-layout tensors, whenever it changes.
"""
from types import SimpleNamespace

import torch

from . import lib as L

# entry points of the model classes that launch kernels or build device layouts: wrapped (once per subclass) so that they
# run under the model's element type
_ELEM_METHODS = ("forward", "forward_tokens", "__call__", "_prepared", "_prepared_encoder", "decode", "decode_tokens",
                 "decode_video", "encode", "time_rows", "release_raw_weights")


def _in_element_type(fn):
    import functools

    @functools.wraps(fn)
    def run(self, *args, **kwargs):
        if L.ELEM[0] is self._elem:
            return fn(self, *args, **kwargs)
        with L.element_type(self._elem):
            return fn(self, *args, **kwargs)
    run._vx_elem_wrapped = True
    return run


def _norm_device(device):
    """torch.device with an explicit index for CUDA ('cuda' == the current device), so that the two spellings of one
    GPU compare equal."""
    device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    if device.type == "cuda" and device.index is None:
        idx = torch.cuda.current_device() if torch.cuda.is_available() else 0
        device = torch.device("cuda", idx)
    return device


class DeviceModule:
    """Base of every model class.  Subclasses keep raw (reference-layout) tensors in `_raw` and build their device
    layouts lazily in `_prepared()`; `expected_keys()` (name -> shape) enables strict loading where it is defined."""

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        for name in _ELEM_METHODS:
            fn = cls.__dict__.get(name)
            if callable(fn) and not getattr(fn, "_vx_elem_wrapped", False):
                setattr(cls, name, _in_element_type(fn))

    def __init__(self):
        self._device = torch.device("cpu")
        self._dtype = torch.bfloat16
        self._raw = {}
        self._P = None
        self._released = False

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return self._dtype

    @property
    def _elem(self):
        """The 16-bit element type the kernels compute on for this model (float32 models: bfloat16 underneath)."""
        return torch.float16 if self._dtype == torch.float16 else torch.bfloat16

    def _invalidate(self):
        self._P = None

    def to(self, *args, **kwargs):
        device = self._device
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, bool) or a is None:          # non_blocking= / copy= flags of torch's signature
                continue
            if isinstance(a, torch.dtype):
                if a not in (torch.bfloat16, torch.float32, torch.float16):
                    raise TypeError(f"unsupported dtype {a}")
                before = self._elem
                self._dtype = a
                if self._elem != before and (self._P is not None or self._released):
                    # the device layouts hold elements of the other type: rebuild them from the source-layout tensors
                    if self._released:
                        raise RuntimeError("release_raw_weights() dropped the source-layout weights; this model can no "
                                           "longer change its element type - load the state dict again first")
                    self._invalidate()
            elif isinstance(a, (torch.device, str, int)):
                device = _norm_device(a)
        if device != _norm_device(self._device):
            if self._released:
                raise RuntimeError("release_raw_weights() dropped the source-layout weights; this model can no longer "
                                   "move to another device - load the state dict again first")
            self._device = device
            self._invalidate()           # device layouts are rebuilt on the new device at the next call
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device) if device is not None else "cuda")

    def half(self):
        return self.to(torch.float16)

    def bfloat16(self):
        return self.to(torch.bfloat16)

    def eval(self):
        return self

    def train(self, mode=False):
        if mode:
            raise NotImplementedError("inference only")
        return self

    def requires_grad_(self, flag=False):
        return self

    # ---- weights
    def expected_keys(self):
        """{name: shape} of the reference module's state_dict, or None when the schema is open (VAE prefixes)."""
        return None

    def load_state_dict(self, state_dict, strict=True):
        expected = self.expected_keys()
        if expected is None:
            self._raw.update({k: v.detach() for k, v in state_dict.items()})
            missing, unexpected = [], []
        else:
            unexpected = [k for k in state_dict if k not in expected]
            for k, v in state_dict.items():
                if k in expected:
                    if tuple(v.shape) != tuple(expected[k]):
                        raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)} from "
                                           f"checkpoint, the shape in current model is {tuple(expected[k])}.")
                    self._raw[k] = v.detach()
            missing = [k for k in expected if k not in state_dict]
            if strict and (missing or unexpected):
                raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: missing keys "
                                   f"{missing[:5]}{'...' if len(missing) > 5 else ''}, unexpected keys "
                                   f"{unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
        self._released = False
        self._invalidate()
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def state_dict(self):
        """The loaded tensors under the reference's key names (what `load_state_dict` received)."""
        if self._released:
            raise RuntimeError("release_raw_weights() dropped the source-layout weights")
        return dict(self._raw)

    def parameters(self):
        return iter(self.state_dict().values())

    def release_raw_weights(self):
        """Drop the source-layout copies once the device layouts exist (frees host/device memory)."""
        self._prepared()
        self._raw = {}
        self._released = True

    def _need_gpu(self):
        if self._device.type != "cuda":
            raise RuntimeError("v_express_amd models run on an MI355X only: call .to('cuda') (no CPU path exists)")
