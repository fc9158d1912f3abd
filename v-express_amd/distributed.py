"""Window x CFG-half work units over the GPUs of one node (one process per GPU, RCCL through torch.distributed).

The reference has no multi-GPU inference (its `--do_multi_devices_inference` flag is never read: inference.py:47,
pipelines/v_express_pipeline.py:433,616,643 — SURVEY.md §0.6).  What makes sharding exact is the loop's own
invariant: within one timestep every window's UNet call reads only the step-start latents
(pipelines/v_express_pipeline.py:526-583), and the two CFG halves of a window are independent batch rows.
So a timestep is 2*W independent units; each rank computes its units, ONE all-gather per timestep exchanges the
raw conv_out predictions (<= 2 MiB per unit), and every rank redundantly applies CFG + mean-overlap + DDIM to the
full clip — sums of <= 2 terms, identical bits on every rank, no all-reduce.

Frame sharding inside a unit ("Ulysses-style"): when a clip has fewer units than ranks (a
16-frame clip is 2 units), S ranks share one unit, each holding f/S frames of the window.  Everything in the UNet is
per-frame except the temporal attention of the 21 motion modules, which needs all f frames of a pixel: there the group
switches layout with one all-to-all ([b, f/S, hw, C] -> [b, f, hw/S, C]), runs the whole temporal transformer on its
pixel slice, and switches back with a second all-to-all before the (per-token) output projection + residual.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class CommTimer:
    """Optional timing of every collective of the data path (bench.py --gpus N: the all-gather of a timestep's
    predictions, the decode's frame gather, the frame-shard all-to-alls, the broadcast of rank 0's noise draw): `with CommTimer() as t:` records a pair of
    events on the current stream around each call; `t.summary()` -> {kind: dict(calls, ms, bytes)}.  Off (no events, no
    overhead) outside the context."""
    active = None

    def __init__(self):
        self.records = []          # (kind, bytes, start_event, end_event)

    def __enter__(self):
        CommTimer.active = self
        return self

    def __exit__(self, *a):
        CommTimer.active = None

    def summary(self):
        if self.records and self.records[0][2] is not None:
            torch.cuda.synchronize()
        out = {}
        for kind, nbytes, s, e in self.records:
            d = out.setdefault(kind, dict(calls=0, ms=0.0, bytes=0))
            d["calls"] += 1
            d["bytes"] += nbytes
            if s is not None:
                d["ms"] += s.elapsed_time(e)
        return out


class _timed_collective:
    def __init__(self, kind, t):
        self.t = CommTimer.active
        self.kind, self.bytes, self.cuda = kind, t.numel() * t.element_size(), t.is_cuda

    def __enter__(self):
        self.s = self.e = None
        if self.t is not None and self.cuda:
            self.s, self.e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *a):
        if self.t is not None:
            if self.e is not None:
                self.e.record()
            self.t.records.append((self.kind, self.bytes, self.s, self.e))


def partition_units(num_windows: int, world_size: int, halves: int = 2) -> List[List[Tuple[int, int]]]:
    """Assignment of the (window, cfg_half) units of one timestep to the ranks (sizes differ by at most one).

    Evenly divisible (or no classifier-free guidance): contiguous blocks of the list (w0,u),(w0,c),(w1,u),... - both
    halves of a window usually land on one rank and run as one b = 2 batch.

    Uneven with CFG (the config-4 clip: 20 units on 8 GPUs = quotas 3,3,3,3,2,2,2,2): every rank holding one unit MORE
    than the others gets whole windows plus a lone UNCONDITIONAL half - the cheaper one (no reference attention, no audio
    cross-attention: 28.2 vs 33.1 ms at 512x512, f = 16; a merged [u, c, u] call 74.0 ms against 78.4 ms for [c, u, c],
    profiles/r02e_host_overhead.json) - and the orphaned conditional halves are paired two by two on the lighter ranks
    in place of one of their windows.  The heaviest rank sets the step time, so this is worth ~6 % at 8 GPUs; results do
    not depend on the assignment (every rank applies the same combine + DDIM update to the gathered predictions).
    halves = 1: no classifier-free guidance (guidance_scale <= 1), a unit is a whole window's single batch row."""
    units = [(w, h) for w in range(num_windows) for h in range(halves)]
    n = len(units)
    base, extra = divmod(n, world_size)

    def contiguous():
        out, pos = [], 0
        for r in range(world_size):
            sz = base + (1 if r < extra else 0)
            out.append(units[pos:pos + sz])
            pos += sz
        return out

    light = world_size - extra
    # balanced form: heavy ranks have an odd quota base + 1 (base even), and enough light ranks can swap one window
    # for two orphaned conditional halves
    if halves != 2 or extra == 0 or base % 2 or base < 2 or extra % 2 or light < extra // 2:
        return contiguous()
    out = [[] for _ in range(world_size)]
    w = 0
    for r in range(extra):                                  # heavy: base / 2 whole windows + the uncond half of one more
        for _ in range(base // 2):
            out[r] += [(w, 0), (w, 1)]
            w += 1
    split = list(range(w, w + extra))                       # the windows whose halves are separated
    for r in range(extra):
        out[r].append((split[r], 0))
    w += extra
    orphans = [(sw, 1) for sw in split]
    for j in range(light):
        r = extra + j
        quota = base
        if orphans:                                         # two orphaned cond halves instead of one window
            out[r] += [orphans.pop(0), orphans.pop(0)]
            quota -= 2
        for _ in range(quota // 2):
            out[r] += [(w, 0), (w, 1)]
            w += 1
    assert w == num_windows and not orphans and sorted(u for a in out for u in a) == units
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


def choose_frame_shards(num_windows: int, world_size: int, window_frames: int, min_hw: int, halves: int = 2) -> int:
    """Largest power-of-two S such that the clip still has too few units for the ranks without it
    (halves*W*S <= world), S divides the world size, the window length and the token count of the coarsest UNet level."""
    s = 1
    while (halves * num_windows * (2 * s) <= world_size and world_size % (2 * s) == 0 and window_frames % (2 * s) == 0
           and min_hw % (2 * s) == 0):
        s *= 2
    return s


def choose_mixed_shards(num_units: int, world_size: int, window_frames: int, min_hw: int) -> int:
    """S in {2, 4} when the units left over after every rank got floor(units / world) whole ones fill the node exactly
    once as S-way frame-sharded units (extra * S == world), else 1.  The config-4 clip: 20 units on 8 GPUs = 2 whole
    units per rank + 4 units left over = one half-unit (8 of the window's 16 frames) per rank, i.e. 2.5 unit-times per
    step on every rank instead of 3 on half of them (whole-unit schedules cannot do better than 20 / 3 = 6.67x)."""
    if world_size < 2 or num_units < world_size:
        return 1
    extra = num_units % world_size
    if extra == 0 or world_size % extra:
        return 1
    s = world_size // extra
    if s not in (2, 4) or window_frames % s or min_hw % s:
        return 1
    return s


class MixedUnitSchedule:
    """Whole units first, the left-over units frame-sharded S ways (`choose_mixed_shards`); identical on every rank.
    Exchange granule = 1/S of a unit (f / S frames): a whole unit fills S consecutive slots of its rank, a sharded unit
    one slot on each of its S ranks; `slots[unit][j]` = (rank, slot) of the unit's j-th frame granule."""

    def __init__(self, num_windows: int, world_size: int, shards: int, halves: int = 2):
        units = [(w, h) for w in range(num_windows) for h in range(halves)]
        base, extra = divmod(len(units), world_size)
        if shards < 2 or extra * shards != world_size or base < 1:
            raise ValueError(f"{len(units)} units on {world_size} ranks do not split into whole units + {shards}-way "
                             "sharded left-overs")
        self.num_windows, self.world_size, self.shards, self.halves = num_windows, world_size, shards, halves
        self.granules = shards
        self.whole = [units[r * base:(r + 1) * base] for r in range(world_size)]
        self.split = units[base * world_size:]                  # unit i lives on ranks [i * S, (i + 1) * S)
        self.max_slots = base * shards + 1
        self.slots = {}
        for r, us in enumerate(self.whole):
            for k, u in enumerate(us):
                self.slots[u] = [(r, k * shards + j) for j in range(shards)]
        for i, u in enumerate(self.split):
            self.slots[u] = [(i * shards + j, base * shards) for j in range(shards)]

    def whole_calls(self, rank: int):
        return group_calls(self.whole[rank])

    def split_unit(self, rank: int):
        return self.split[rank // self.shards]

    def rounds(self) -> float:
        """Unit-times on the critical path of one timestep."""
        return len(self.whole[0]) + 1.0 / self.shards


class UnitSchedule:
    """Static (window, cfg_half) -> (rank group, slot) map of one clip; identical on every rank.  With
    frame_shards = S > 1 the ranks form world/S groups of S consecutive ranks; a group owns units like a single rank
    does for S = 1 and member j of the group computes frames [j*f/S, (j+1)*f/S) of each of them."""

    def __init__(self, num_windows: int, world_size: int, frame_shards: int = 1, halves: int = 2):
        if frame_shards < 1 or world_size % frame_shards:
            raise ValueError(f"frame_shards={frame_shards} must divide the world size {world_size}")
        self.num_windows, self.world_size, self.frame_shards = num_windows, world_size, frame_shards
        self.halves = halves
        self.groups = world_size // frame_shards
        self.assign = partition_units(num_windows, self.groups, halves)
        self.max_units = max(len(a) for a in self.assign)
        self.slot = {}
        for r, units in enumerate(self.assign):
            for s, u in enumerate(units):
                self.slot[u] = (r, s)

    def calls(self, rank: int):
        return group_calls(self.assign[rank // self.frame_shards])

    def unit_ranks(self, unit):
        """(ranks holding the frame shards of `unit`, in frame order), slot."""
        g, s = self.slot[unit]
        return [g * self.frame_shards + j for j in range(self.frame_shards)], s

    def rounds(self) -> int:
        """Half-window UNet passes on the critical path of one timestep (ideal speed-up = 2W / rounds)."""
        return self.max_units


@dataclass
class DistContext:
    rank: int = 0
    world_size: int = 1
    group: Optional[object] = None
    force: bool = False       # tools/rccl_world1_probe.py: run the collective code path on a one-rank group as well

    @property
    def enabled(self):
        return self.world_size > 1 or self.force

    @property
    def backend(self):
        """Collective backend of the group ("nccl" = RCCL over xGMI, the only measured configuration; anything else
        stages CUDA tensors through the host and exists for control-flow tests on one GPU)."""
        return dist.get_backend(self.group) if self.enabled else None

    @staticmethod
    def from_env():
        if dist.is_available() and dist.is_initialized():
            return DistContext(dist.get_rank(), dist.get_world_size(), None)
        return DistContext()

    def _all_gather(self, local: torch.Tensor) -> torch.Tensor:
        out = torch.empty((self.world_size,) + tuple(local.shape), device=local.device, dtype=local.dtype)
        with _timed_collective("all_gather", out):
            if local.is_cuda and dist.get_backend(self.group) != "nccl":
                # control-flow testing on a single-GPU box (gloo): stage through the host; never the measured path
                host = torch.empty(out.shape, dtype=local.dtype)
                dist.all_gather_into_tensor(host.view(-1), local.detach().cpu().contiguous().view(-1), group=self.group)
                out.copy_(host)
            else:
                dist.all_gather_into_tensor(out.view(-1), local.contiguous().view(-1), group=self.group)
        return out

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        """In-place broadcast from `src` (rank 0's noise draw becomes every rank's initial latents)."""
        if not self.enabled:
            return t
        with _timed_collective("broadcast", t):           # (timed like every other collective of the data path)
            if t.is_cuda and dist.get_backend(self.group) != "nccl":
                host = t.detach().cpu().contiguous()          # single-GPU control-flow testing over gloo
                dist.broadcast(host, src=src, group=self.group)
                t.copy_(host)
            else:
                dist.broadcast(t, src=src, group=self.group)
        return t

    def all_gather_units(self, local: torch.Tensor, max_units: int) -> torch.Tensor:
        """local: [max_units, ...] (this rank's unit outputs, zero-padded) -> [world, max_units, ...]."""
        if not self.enabled:
            return local.unsqueeze(0)
        return self._all_gather(local)

    def all_gather_frames(self, local: torch.Tensor) -> torch.Tensor:
        if not self.enabled:
            return local.unsqueeze(0)
        return self._all_gather(local)

    def frame_shard(self, frame_shards: int) -> Optional["FrameShard"]:
        """This rank's FrameShard for groups of `frame_shards` consecutive ranks (None for 1).  Creating the process
        groups is collective: every rank must call this with the same value (the pipeline does, once per clip)."""
        if frame_shards <= 1:
            return None
        if self.world_size % frame_shards:
            raise ValueError(f"frame_shards={frame_shards} must divide the world size {self.world_size}")
        cache = self.__dict__.setdefault("_shard_groups", {})
        if frame_shards not in cache:
            mine = None
            for g in range(self.world_size // frame_shards):
                ranks = list(range(g * frame_shards, (g + 1) * frame_shards))
                grp = dist.new_group(ranks)
                if self.rank in ranks:
                    mine = grp
            cache[frame_shards] = mine
        return FrameShard(self.rank % frame_shards, frame_shards, cache[frame_shards])


class FrameShard:
    """Layout switch of one frame-sharded unit between its S ranks (one all-to-all each way).
    frame shard:  [b * f/S, hw, C]   - this rank's frames of every batch row, all pixels
    pixel shard:  [b * f, hw/S, C]   - all frames, this rank's pixel slice."""

    def __init__(self, index: int, size: int, group=None):
        self.index, self.size, self.group = index, size, group

    def _all_to_all(self, send: torch.Tensor) -> torch.Tensor:
        """send[r] goes to member r; the result's [s] came from member s."""
        recv = torch.empty_like(send)
        with _timed_collective("all_to_all", send):
            if send.is_cuda and dist.get_backend(self.group) != "nccl":
                # control-flow testing on a single-GPU box (gloo): stage through the host; never the measured path
                hs = send.cpu()
                hr = torch.empty_like(hs)
                dist.all_to_all_single(hr, hs, group=self.group)
                recv.copy_(hr)
            else:
                dist.all_to_all_single(recv, send, group=self.group)
        return recv

    def to_pixel_shard(self, x: torch.Tensor, b: int, f_loc: int) -> torch.Tensor:
        n, hw, c = x.shape
        S = self.size
        if n != b * f_loc or hw % S:
            raise ValueError(f"frame shard [{n}, {hw}, {c}] does not split over {S} ranks (b={b}, f/S={f_loc})")
        send = x.view(n, S, hw // S, c).transpose(0, 1).contiguous()                   # [S(dst), b*f_loc, hw/S, C]
        recv = self._all_to_all(send)                                                    # [S(src), b*f_loc, hw/S, C]
        return recv.view(S, b, f_loc, hw // S, c).permute(1, 0, 2, 3, 4).reshape(b * S * f_loc, hw // S, c)

    def to_frame_shard(self, h: torch.Tensor, b: int, f_loc: int) -> torch.Tensor:
        n, hw_t, c = h.shape
        S = self.size
        if n != b * S * f_loc:
            raise ValueError(f"pixel shard [{n}, {hw_t}, {c}] is not b*f = {b}*{S * f_loc} frames")
        send = h.view(b, S, f_loc, hw_t, c).permute(1, 0, 2, 3, 4).contiguous()         # [S(dst), b, f_loc, hw/S, C]
        recv = self._all_to_all(send)                                                    # [S(src = pixel slice), ...]
        return recv.view(S, b * f_loc, hw_t, c).transpose(0, 1).reshape(b * f_loc, S * hw_t, c)


def split_frames(num_frames: int, world_size: int) -> List[Tuple[int, int]]:
    """Even contiguous split of the decode work: rank r decodes frames [lo, hi)."""
    per = -(-num_frames // world_size)
    return [(min(num_frames, r * per), min(num_frames, (r + 1) * per)) for r in range(world_size)]
