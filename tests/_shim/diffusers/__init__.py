"""Minimal stand-in for diffusers==0.29.2 — TEST INFRASTRUCTURE ONLY.

diffusers is not installed in this image and cannot be installed offline, yet every
file under /root/reference/modules and /root/reference/pipelines imports it.  This
package restates, from the published 0.29.2 behaviour, only the leaf classes those
files use, so the reference's own block wiring,
monkey-patching and window loop can be imported UNMODIFIED and used as the Tier-A
oracle that pins `oracle/` (Tier B).  Nothing in the product path imports this.

It is put first on sys.path by tests/ref_import.py, and only in this container
(/root/reference does not exist on the GPU box).
"""
import sys
import types

from . import _impl as I


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], child, m)
    return m


_stub = I.make_stub

_mod("diffusers.configuration_utils", ConfigMixin=I.ConfigMixin, register_to_config=I.register_to_config)
_mod("diffusers.loaders", UNet2DConditionLoadersMixin=I.UNet2DConditionLoadersMixin)
_mod("diffusers.utils",
     BaseOutput=I.BaseOutput, logging=I.logging, deprecate=I.deprecate, is_torch_version=I.is_torch_version,
     USE_PEFT_BACKEND=True, scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None,
     is_accelerate_available=lambda: False, SAFETENSORS_WEIGHTS_NAME="diffusion_pytorch_model.safetensors",
     WEIGHTS_NAME="diffusion_pytorch_model.bin")
_mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
_mod("diffusers.utils.torch_utils", randn_tensor=I.randn_tensor, apply_freeu=_stub("apply_freeu"))
_mod("diffusers.models", ModelMixin=I.ModelMixin)
_mod("diffusers.models.modeling_utils", ModelMixin=I.ModelMixin)
_mod("diffusers.models.activations", get_activation=I.get_activation)
_mod("diffusers.models.attention_processor",
     Attention=I.Attention, AttnProcessor=I.AttnProcessor, AttnProcessor2_0=I.AttnProcessor2_0,
     AttentionProcessor=object, AttnAddedKVProcessor=_stub("AttnAddedKVProcessor"),
     ADDED_KV_ATTENTION_PROCESSORS=(), CROSS_ATTENTION_PROCESSORS=())
_mod("diffusers.models.attention",
     Attention=I.Attention, FeedForward=I.FeedForward, GEGLU=I.GEGLU,
     AdaLayerNorm=_stub("AdaLayerNorm"), AdaLayerNormZero=_stub("AdaLayerNormZero"),
     GatedSelfAttentionDense=_stub("GatedSelfAttentionDense"))
_mod("diffusers.models.embeddings",
     Timesteps=I.Timesteps, TimestepEmbedding=I.TimestepEmbedding,
     SinusoidalPositionalEmbedding=_stub("SinusoidalPositionalEmbedding"),
     CaptionProjection=_stub("CaptionProjection"),
     PixArtAlphaTextProjection=_stub("PixArtAlphaTextProjection"),
     GaussianFourierProjection=_stub("GaussianFourierProjection"),
     ImageHintTimeEmbedding=_stub("ImageHintTimeEmbedding"), ImageProjection=_stub("ImageProjection"),
     ImageTimeEmbedding=_stub("ImageTimeEmbedding"), TextImageProjection=_stub("TextImageProjection"),
     TextImageTimeEmbedding=_stub("TextImageTimeEmbedding"), TextTimeEmbedding=_stub("TextTimeEmbedding"),
     PositionNet=_stub("PositionNet"), GLIGENTextBoundingboxProjection=_stub("GLIGENTextBoundingboxProjection"))
_mod("diffusers.models.normalization", AdaLayerNormSingle=_stub("AdaLayerNormSingle"))
_mod("diffusers.models.lora", LoRACompatibleConv=I.nn.Conv2d, LoRACompatibleLinear=I.nn.Linear)
_mod("diffusers.models.resnet", ResnetBlock2D=I.ResnetBlock2D, Downsample2D=I.Downsample2D, Upsample2D=I.Upsample2D)
_mod("diffusers.models.transformers")
_mod("diffusers.models.transformers.dual_transformer_2d", DualTransformer2DModel=_stub("DualTransformer2DModel"))
_mod("diffusers.image_processor", VaeImageProcessor=I.VaeImageProcessor)
_mod("diffusers.schedulers",
     DDIMScheduler=I.DDIMScheduler, DPMSolverMultistepScheduler=_stub("DPMSolverMultistepScheduler"),
     EulerAncestralDiscreteScheduler=_stub("EulerAncestralDiscreteScheduler"),
     EulerDiscreteScheduler=_stub("EulerDiscreteScheduler"),
     LMSDiscreteScheduler=_stub("LMSDiscreteScheduler"), PNDMScheduler=_stub("PNDMScheduler"))

DiffusionPipeline = I.DiffusionPipeline
AutoencoderKL = I.AutoencoderKL
DDIMScheduler = I.DDIMScheduler
__version__ = "0.29.2+shim"
