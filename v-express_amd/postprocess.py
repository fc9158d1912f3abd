"""Post-processing right after the decode: the 3x3x3 temporal-spatial median filter and the
uint8 frame packing of `save_video` (pipelines/utils.py:46-73) as one HBM-bound kernel (`vx_median3d`).  The reference
pads on the host, hops every frame segment to the device, unfolds 27 shifted copies and calls torch.median, then
multiplies by 255 and casts on the host; here the clip never leaves the device until the packed frames are needed.
Video muxing (cv2 / ffmpeg, pipelines/utils.py:75-87) stays outside."""
import torch

from . import ops


def median_filter_3d(video_tensor, kernel_size=3, device=None):
    """Same signature and result as pipelines/utils.py:median_filter_3d: `[C, F, H, W]` float -> `[C, F, H, W]` float32
    (bit-identical: the median is a selection).  `device` is where the filter runs (default: the tensor's device,
    or cuda:0 for a host tensor); the result comes back on the input's device like the reference's."""
    if kernel_size != 3:
        raise NotImplementedError("V-Express filters with kernel_size=3 (pipelines/utils.py:70)")
    src = video_tensor.device
    dev = torch.device(device) if device is not None else (src if src.type == "cuda" else torch.device("cuda", 0))
    out, _ = ops.median3d(video_tensor.to(dev, torch.float32).contiguous(), want_f32=True)
    return out.to(src)


def video_frames_uint8(video, median=True):
    """`video` `[1, 3, F, H, W]` (or `[3, F, H, W]`) float in [0, 1] on the device -> uint8 `[F, H, W, 3]` frames,
    median-filtered first like `save_video` (pipelines/utils.py:64-73)."""
    v = video[0] if video.dim() == 5 else video
    v = v.to(torch.float32).contiguous()
    if median:
        _, u8 = ops.median3d(v, want_f32=False, want_u8=True)
        return u8
    return (v.permute(1, 2, 3, 0) * 255).to(torch.uint8)
