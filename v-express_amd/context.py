"""Sliding-window context scheduler + the host-side plan of the mean-overlap loop.

`uniform` / `ordered_halving` / `get_context_scheduler` are a near-verbatim TRANSCRIPTION of pipelines/context.py:22-66
(~25 lines of integer index arithmetic, same generator signature): the window lists are part of the numerical
contract - any other formulation would have to reproduce them bit for bit - so the arithmetic is kept as it is there
and asserted identical to the reference module's output (tests/golden/windows.pt, test_oracle_vs_reference.py).
Everything else in this file is original.  `overlap_plan` turns the per-window bookkeeping of pipelines/v_express_pipeline.py:498-500,552-572
into a static table the device kernels consume: for every frame, which (window, position) predictions make up
its averaged noise prediction and by what count they are divided — including the reference's behaviour for a
reflected last window with duplicate frame ids: the count is incremented once, the
duplicated frame is stepped twice and the LAST write wins.
"""
from typing import Callable, List

import numpy as np


def ordered_halving(val):
    bin_str = f"{val:064b}"
    return int(bin_str[::-1], 2) / (1 << 64)


def uniform(step: int = ..., num_frames: int = ..., context_size: int = None, context_stride: int = 3,
            context_overlap: int = 4, closed_loop: bool = True):
    if num_frames <= context_size:
        yield list(range(num_frames))
        return
    context_stride = min(context_stride, int(np.ceil(np.log2(num_frames / context_size))) + 1)
    for context_step in 1 << np.arange(context_stride):
        context_step = int(context_step)
        pad = int(round(num_frames * ordered_halving(step)))
        for j in range(int(ordered_halving(step) * context_step) + pad,
                       num_frames + pad + (0 if closed_loop else -context_overlap),
                       (context_size * context_step - context_overlap)):
            next_itr = []
            for e in range(j, j + context_size * context_step, context_step):
                if e >= num_frames:
                    e = num_frames - 2 - e % num_frames
                next_itr.append(e)
            yield next_itr


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
