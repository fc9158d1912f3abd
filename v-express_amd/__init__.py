"""v-express_amd — MI355X-native V-Express denoising hot path.

Drop-in for the reference's `VExpressPipeline.__call__` / `UNet3DConditionModel.forward` hot path
(pipelines/v_express_pipeline.py:409-646, modules/unet_3d.py:400-578 in tencent-ailab/V-Express),
executed by hand-written gfx950 HIP kernels behind the C ABI declared in include/vexpress_hip.h.
There is no CPU/PyTorch fallback: importing the model classes loads libvexpress_hip.so and raises if it
is missing.  Host-only helpers (synth, context, scheduler, distributed) import without the library.
"""
from .synth import UNetConfig, VaeConfig  # noqa: F401
from .scheduler import DDIMScheduler  # noqa: F401

_LAZY = {
    "UNet3DConditionModel": "unet_3d", "UNet3DConditionOutput": "unet_3d", "UNet2DConditionModel": "unet_2d",
    "AutoencoderKLDecoder": "vae", "AutoencoderKL": "vae", "ReferenceAttentionControl": "mutual_self_attention",
    "VExpressPipeline": "pipeline", "VKpsGuider": "prologue", "AudioProjection": "prologue",
    "median_filter_3d": "postprocess", "video_frames_uint8": "postprocess",
    "Wav2Vec2Model": "wav2vec2", "WaveformProcessor": "wav2vec2",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(f"{__name__}.{_LAZY[name]}"), name)
    raise AttributeError(name)


__all__ = ["UNetConfig", "VaeConfig", "DDIMScheduler"] + list(_LAZY)
