"""Architecture constants of the oracle (TEST INFRASTRUCTURE).

Values follow /root/reference/inference_v2.yaml:1-23 and the SD-1.5 UNet config the reference
loads (`model_ckpts/stable-diffusion-v1-5/unet/config.json`).
"""

from dataclasses import dataclass
from typing import Tuple


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    heads: int = 8                      # `attention_head_dim: 8` is used as the head COUNT (unet_3d_blocks.py:353)
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5              # resnet / conv_norm_out GroupNorm
    temporal_max_len: int = 32          # inference_v2.yaml:21
    attn_levels: Tuple[bool, ...] = (True, True, True, False)   # CrossAttn{Down,Up}Block vs plain

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4


@dataclass
class VaeConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215
