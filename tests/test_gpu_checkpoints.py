"""Checkpoint ingestion ON THE DEVICE: see tests/ckpt_cases.py - synthetic
checkpoint files in the reference's formats loaded through `checkpoints.load_*` onto the GPU in the reference's load
order (inference.py:77-129); every loaded model's forward equals the `load_state_dict` model bit for bit."""
import pytest
import torch

import ckpt_cases

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("layout", ["new_attn", "old_attn"])
def test_denoising_unet_and_reference_net_from_files(tmp_path, layout):
    _need_gpu()
    ckpt_cases.unet_and_reference_net_from_files(tmp_path, layout, "cuda")


def test_vae_guider_and_audio_projection_from_files(tmp_path):
    _need_gpu()
    ckpt_cases.vae_guider_and_audio_projection_from_files(tmp_path, "cuda")
