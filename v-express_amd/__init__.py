"""v-express_amd — MI355X-native V-Express denoising hot path.

Drop-in for the reference's `VExpressPipeline.__call__` / `UNet3DConditionModel.forward` hot path
(pipelines/v_express_pipeline.py:409-646, modules/unet_3d.py:400-578 in tencent-ailab/V-Express),
executed by hand-written gfx950 HIP kernels behind the C ABI declared in include/vexpress_hip.h.
There is no CPU/PyTorch fallback: importing `v_express_amd.lib` raises if the HIP library is missing.
"""
from .synth import UNetConfig, VaeConfig  # noqa: F401

__all__ = ["UNetConfig", "VaeConfig"]
