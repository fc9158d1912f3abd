"""Import /root/reference UNMODIFIED over the diffusers stand-in (Tier-A oracle; dev container only).

TEST INFRASTRUCTURE.  `have_reference()` is False on the GPU box, where /root/reference does not exist;
every test that needs it must skip there and rely on tests/golden/ instead.
"""
import os
import sys

REFERENCE_ROOT = "/root/reference"
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shim")


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "modules"))


def import_reference():
    """Returns (modules, pipelines) packages of the reference, imported as-is."""
    if not have_reference():
        raise RuntimeError("/root/reference is not available here")
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    import diffusers  # noqa: F401  (the stand-in)
    import modules
    import pipelines
    return modules, pipelines


def import_reference_utils():
    """pipelines/utils.py (median_filter_3d, save_video).  Its module-level imports of cv2 / imageio_ffmpeg (video
    muxing only, not installed here) are satisfied with empty stand-in modules; median_filter_3d uses neither."""
    import types
    import_reference()
    for name in ("cv2", "imageio_ffmpeg"):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                m = types.ModuleType(name)
                m.get_ffmpeg_exe = lambda: "ffmpeg"
                sys.modules[name] = m
    import pipelines.utils as U
    return U


SD15_UNET_CONFIG = dict(
    sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    mid_block_type="UNetMidBlock2DCrossAttn",
    up_block_types=["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    only_cross_attention=False, block_out_channels=[320, 640, 1280, 1280], layers_per_block=2,
    downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
    cross_attention_dim=768, attention_head_dim=8, use_linear_projection=False,
)

UNET_ADDITIONAL_KWARGS = dict(
    use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
    use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=True,
    motion_module_decoder_only=False, motion_module_type="Vanilla",
    motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                              attention_block_types=["Temporal_Self", "Temporal_Self"],
                              temporal_position_encoding=True, temporal_position_encoding_max_len=32,
                              temporal_attention_dim_div=1),
)

NOISE_SCHEDULER_KWARGS = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                              steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True,
                              timestep_spacing="trailing")
