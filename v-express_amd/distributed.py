"""Window x CFG-half work units over the GPUs of one node (one process per GPU, RCCL through torch.distributed).

The reference has no multi-GPU inference (its `--do_multi_devices_inference` flag is never read: inference.py:47,
pipelines/v_express_pipeline.py:433,616,643 — SURVEY.md §0.6).  What makes sharding exact is the loop's own
invariant: within one timestep every window's UNet call reads only the step-start latents
(pipelines/v_express_pipeline.py:526-583), and the two CFG halves of a window are independent batch rows.
So a timestep is 2*W independent units; each rank computes its units, ONE all-gather per timestep exchanges the
raw conv_out predictions (<= 2 MiB per unit), and every rank redundantly applies CFG + mean-overlap + DDIM to the
full clip — sums of <= 2 terms, identical bits on every rank, no all-reduce.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def partition_units(num_windows: int, world_size: int) -> List[List[Tuple[int, int]]]:
    """Contiguous block assignment of the (window, cfg_half) units: rank r gets units [starts[r], starts[r+1])
    of the list (w0,u),(w0,c),(w1,u),(w1,c),...  Sizes differ by at most one, adjacent units stay together so
    that both halves of a window usually land on one rank and run as one b=2 batch."""
    units = [(w, h) for w in range(num_windows) for h in range(2)]
    n = len(units)
    base, extra = divmod(n, world_size)
    out, pos = [], 0
    for r in range(world_size):
        sz = base + (1 if r < extra else 0)
        out.append(units[pos:pos + sz])
        pos += sz
    return out


def group_calls(units: List[Tuple[int, int]]) -> List[Tuple[int, List[int]]]:
    """Merge a rank's units into UNet calls: [(window, [halves])] with halves == [0,1] when both are local."""
    calls = []
    for w, h in units:
        if calls and calls[-1][0] == w:
            calls[-1][1].append(h)
        else:
            calls.append((w, [h]))
    return calls


class UnitSchedule:
    """Static (window, cfg_half) -> (rank, slot) map of one clip; identical on every rank."""

    def __init__(self, num_windows: int, world_size: int):
        self.num_windows, self.world_size = num_windows, world_size
        self.assign = partition_units(num_windows, world_size)
        self.max_units = max(len(a) for a in self.assign)
        self.slot = {}
        for r, units in enumerate(self.assign):
            for s, u in enumerate(units):
                self.slot[u] = (r, s)

    def calls(self, rank: int):
        return group_calls(self.assign[rank])

    def rounds(self) -> int:
        """Half-window UNet passes on the critical path of one timestep (ideal speed-up = 2W / rounds)."""
        return self.max_units


@dataclass
class DistContext:
    rank: int = 0
    world_size: int = 1
    group: Optional[object] = None

    @property
    def enabled(self):
        return self.world_size > 1

    @staticmethod
    def from_env():
        if dist.is_available() and dist.is_initialized():
            return DistContext(dist.get_rank(), dist.get_world_size(), None)
        return DistContext()

    def _all_gather(self, local: torch.Tensor) -> torch.Tensor:
        out = torch.empty((self.world_size,) + tuple(local.shape), device=local.device, dtype=local.dtype)
        if local.is_cuda and dist.get_backend(self.group) != "nccl":
            # control-flow testing on a single-GPU box (gloo): stage through the host; never the measured path
            host = torch.empty(out.shape, dtype=local.dtype)
            dist.all_gather_into_tensor(host.view(-1), local.detach().cpu().contiguous().view(-1), group=self.group)
            out.copy_(host)
            return out
        dist.all_gather_into_tensor(out.view(-1), local.contiguous().view(-1), group=self.group)
        return out

    def all_gather_units(self, local: torch.Tensor, max_units: int) -> torch.Tensor:
        """local: [max_units, ...] (this rank's unit outputs, zero-padded) -> [world, max_units, ...]."""
        if not self.enabled:
            return local.unsqueeze(0)
        return self._all_gather(local)

    def all_gather_frames(self, local: torch.Tensor) -> torch.Tensor:
        if not self.enabled:
            return local.unsqueeze(0)
        return self._all_gather(local)


def split_frames(num_frames: int, world_size: int) -> List[Tuple[int, int]]:
    """Even contiguous split of the decode work: rank r decodes frames [lo, hi)."""
    per = -(-num_frames // world_size)
    return [(min(num_frames, r * per), min(num_frames, (r + 1) * per)) for r in range(world_size)]
