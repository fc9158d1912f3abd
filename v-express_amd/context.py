"""Sliding-window context scheduler + the host-side plan of the mean-overlap loop.

`uniform` reproduces the window lists of pipelines/context.py:22-60 (the AnimateDiff "uniform" schedule) - they are part
of the numerical contract, bit for bit - in its own formulation (round 3; rounds 1-2 carried a transcription): the
schedule is a set of dilation levels d = 1, 2, 4, ...; on each level windows of `context_size` frames spaced d apart
start every `context_size * d - context_overlap` frames from an offset given by the base-2 radical inverse of `step`,
and a frame index that runs past the clip is reflected to `F - 2 - (e mod F)`.  The pipeline only ever asks for
step = 0, context_stride = 1, closed_loop = False (pipelines/v_express_pipeline.py:471-481): one level, starts
0, s, 2s, ... with s = context_size - context_overlap while start < F - context_overlap.  Asserted identical to the
reference module for a grid of parameters (tests/golden/windows.pt, tests/test_oracle_vs_reference.py).
`overlap_plan` turns the per-window bookkeeping of pipelines/v_express_pipeline.py:498-500,552-572
into a static table the device kernels consume: for every frame, which (window, position) predictions make up
its averaged noise prediction and by what count they are divided - including the reference's behaviour for a
reflected last window with duplicate frame ids: the count is incremented once, the
duplicated frame is stepped twice and the LAST write wins.
"""
from typing import Callable, List

import numpy as np


def radical_inverse_base2(val: int) -> float:
    """The 64-bit bit-reversal of `val` as a fraction in [0, 1) (van der Corput sequence): 0, 1/2, 1/4, 3/4, ..."""
    rev = 0
    for _ in range(64):
        rev = (rev << 1) | (val & 1)
        val >>= 1
    return rev / float(1 << 64)


ordered_halving = radical_inverse_base2          # the reference's name for it (pipelines/context.py:14-19)


def uniform(step: int = ..., num_frames: int = ..., context_size: int = None, context_stride: int = 3,
            context_overlap: int = 4, closed_loop: bool = True):
    """Generator of frame-index lists, one per context window (see the module docstring)."""
    F = num_frames
    if F <= context_size:
        yield list(range(F))
        return
    phase = radical_inverse_base2(step)
    shift = int(round(F * phase))
    levels = min(context_stride, int(np.ceil(np.log2(F / context_size))) + 1)
    last_start = F + shift - (0 if closed_loop else context_overlap)       # exclusive bound of the window starts
    for level in range(levels):
        d = 1 << level
        first = int(phase * d) + shift
        hop = context_size * d - context_overlap
        for start in range(first, last_start, hop):
            frames = start + d * np.arange(context_size)
            frames = np.where(frames >= F, F - 2 - frames % F, frames)     # reflection past the end of the clip
            yield [int(e) for e in frames]


def get_context_scheduler(name: str) -> Callable:
    if name == "uniform":
        return uniform
    raise ValueError(f"Unknown context_overlap policy {name}")


def compute_num_context(init_video_length, context_size, context_overlap):
    """pipelines/context.py:7-11."""
    return (init_video_length - context_size) // (context_size - context_overlap) + 1


def aligned_video_length(init_video_length, context_size, context_overlap):
    """inference.py:255-264: the clip length inference.py actually requests (whole windows only)."""
    n = compute_num_context(init_video_length, context_size, context_overlap)
    return (n - 1) * (context_size - context_overlap) + context_size


def overlap_plan(windows: List[List[int]], num_frames: int):
    """Static plan of one timestep of the mean-overlap loop.

    Returns dict(counts[F], terms: {frame: [(window, latent_idx), ...]}, step_frames: ordered frame list,
    max_terms).  `terms[frame]` are the predictions summed (each divided by counts[frame]) into the value the
    frame's DDIM step finally keeps, replaying v_express_pipeline.py:552-572 symbolically."""
    counts = np.zeros(num_frames, dtype=np.int64)
    for w in windows:
        counts[np.unique(np.asarray(w))] += 1          # tensor index_put: duplicates count once (:498-500)
    counter = np.zeros(num_frames, dtype=np.int64)
    pending = [None] * num_frames
    final = {}
    for wi, w in enumerate(windows):
        counter[np.unique(np.asarray(w))] += 1         # :552
        for li, fi in enumerate(w):                     # :556-564
            if pending[fi] is None:
                pending[fi] = [(wi, li)]
            else:
                pending[fi] = pending[fi] + [(wi, li)]
            if counter[fi] == counts[fi]:
                final[fi] = pending[fi]                 # stepped now; a later duplicate overwrites (:572)
                pending[fi] = None
    leftover = [i for i in range(num_frames) if i not in final]
    if leftover:
        raise ValueError(f"frames {leftover[:8]} are never completed by the window schedule")
    step_frames = sorted(final)
    return dict(counts=counts, terms=final, step_frames=step_frames,
                max_terms=max(len(v) for v in final.values()))
