"""world_size=2 `gloo` test (CPU) of the window x CFG-half sharding: every rank computes only its units, one
all-gather per timestep, redundant combine + DDIM on every rank -> bit-identical to the single-process loop."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import loop as OL
from v_express_amd import context, distributed
from v_express_amd.scheduler import DDIMScheduler

SCHED_KW = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, steps_offset=1,
                prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")


def fake_unit(latents, window, half, t):
    """Stand-in for one CFG half of a UNet call: any deterministic function of the step-start latents."""
    x = latents[0][:, window]                                  # [4, f, h, w]
    return torch.tanh(x * (1.0 + 0.25 * half) + 0.001 * t + x.roll(1, dims=1) * 0.3)


def fake_unit_sharded(latents, window, lo, f_loc, half, t, shard):
    """fake_unit on this rank's f_loc frames of the window; the frame roll (the stand-in for temporal attention)
    runs in the pixel-shard layout between the two all-to-alls, like the motion modules do."""
    x = latents[0][:, window[lo:lo + f_loc]]                    # [4, f_loc, h, w]
    c, _, h, w = x.shape
    tok = x.permute(1, 2, 3, 0).reshape(f_loc, h * w, c)       # [b*f_loc, hw, C] with b = 1
    pix = shard.to_pixel_shard(tok, 1, f_loc)                   # [f, hw/S, C]
    rolled = shard.to_frame_shard(pix.roll(1, dims=0), 1, f_loc)
    rolled = rolled.reshape(f_loc, h, w, c).permute(3, 0, 1, 2)
    return torch.tanh(x * (1.0 + 0.25 * half) + 0.001 * t + rolled * 0.3)


def run_loop(rank, world, F, cs, co, steps, dc, S=1):
    windows = OL.uniform_windows(F, cs, co)
    plan = context.overlap_plan(windows, F)
    f = len(windows[0])
    sch = distributed.UnitSchedule(len(windows), world, S)
    shard = dc.frame_shard(S)
    f_loc, lo = f // S, (rank % S) * (f // S)
    s = DDIMScheduler(**SCHED_KW)
    s.set_timesteps(steps)
    lat = torch.randn(1, 4, F, 4, 4, generator=torch.Generator().manual_seed(0))
    for t in s.timesteps.tolist():
        local = torch.zeros(sch.max_units, 4, f_loc, 4, 4)
        for w, halves in sch.calls(rank):
            for hlf in halves:
                if shard is None:
                    local[sch.slot[(w, hlf)][1]] = fake_unit(lat, windows[w], hlf, t)
                else:
                    local[sch.slot[(w, hlf)][1]] = fake_unit_sharded(lat, windows[w], lo, f_loc, hlf, t, shard)
        gathered = dc.all_gather_units(local, sch.max_units)
        full = {}
        for wi in range(len(windows)):
            for hlf in range(2):
                ranks, slot = sch.unit_ranks((wi, hlf))
                full[(wi, hlf)] = torch.cat([gathered[r, slot] for r in ranks], dim=1)      # frame shards in order
        sa, s1a, sap, s1ap = s.step_coefficients(t)
        new = lat.clone()
        for fr in plan["step_frames"]:
            v = None
            for (wi, li) in plan["terms"][fr]:
                u = full[(wi, 0)][:, li]
                c = full[(wi, 1)][:, li]
                term = (u + 3.5 * (c - u)) / float(plan["counts"][fr])
                v = term if v is None else v + term
            x = lat[0, :, fr]
            new[0, :, fr] = sap * (sa * x - s1a * v) + s1ap * (sa * v + s1a * x)
        lat = new
    return lat


def _worker(rank, world, port, F, cs, co, steps, q, S=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dc = distributed.DistContext.from_env()
    assert dc.enabled and dc.world_size == world and dc.rank == rank
    out = run_loop(rank, world, F, cs, co, steps, dc, S)
    lo, hi = distributed.split_frames(F, world)[rank]
    frames = torch.zeros(distributed.split_frames(F, world)[0][1], 3)
    frames[:hi - lo] = float(rank + 1)
    allf = dc.all_gather_frames(frames).reshape(-1, 3)[:F]
    # by VALUE (numpy arrays are pickled into the queue): a torch tensor travels as a file descriptor that the parent must
    # fetch from THIS process while it is still alive - on a busy machine the child was gone first (EOFError in q.get)
    q.put((rank, out.numpy().copy(), allf[:, 0].numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("F,cs,co", [(28, 8, 2), (11, 4, 2)])
def test_sharded_loop_is_bit_identical_to_single_process(F, cs, co):
    steps, world = 3, 2
    ref = run_loop(0, 1, F, cs, co, steps, distributed.DistContext())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, F, cs, co, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    results = [(r, torch.from_numpy(o), torch.from_numpy(w)) for r, o, w in results]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, owner in results:
        assert torch.equal(out, ref), f"rank {rank} diverged from the single-process loop"
        lo, hi = distributed.split_frames(F, world)[0]
        assert (owner[:hi] == 1).all() and (owner[hi:] == 2).all()


def _spawn(world, F, cs, co, steps, S):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, F, cs, co, steps, q, S)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    results = [(r, torch.from_numpy(o), torch.from_numpy(w)) for r, o, w in results]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("world,S,F,cs,co", [(2, 2, 8, 8, 2), (4, 2, 8, 8, 2), (4, 2, 14, 8, 2), (4, 4, 8, 8, 2),
                                             (4, 1, 32, 8, 2)])      # last: 5 windows on 4 ranks, balanced assignment
def test_frame_sharded_units_are_bit_identical_to_single_process(world, S, F, cs, co):
    """S ranks share each (window, CFG-half) unit, each holding f/S frames; the temporal mixing
    runs in the pixel-shard layout between two all-to-alls inside the unit's process group (world/S groups work on
    different units concurrently).  Data movement only => bit-identical to the single-process loop."""
    ref = run_loop(0, 1, F, cs, co, 2, distributed.DistContext())
    for rank, out, _ in _spawn(world, F, cs, co, 2, S):
        assert torch.equal(out, ref), f"rank {rank} diverged from the single-process loop"


def test_balanced_unit_assignment_for_uneven_cfg_clips():
    """partition_units: every unit exactly once, sizes differ by at most one; on the config-4 clip (10 windows, 8 ranks)
    the 3-unit ranks hold a whole window plus a lone UNCONDITIONAL half, the orphaned conditional halves sit in pairs on
    2-unit ranks; where the balanced form does not apply the blocks stay contiguous."""
    D = distributed
    for W in range(1, 14):
        for R in (1, 2, 3, 4, 6, 8):
            for halves in (1, 2):
                a = D.partition_units(W, R, halves)
                assert sorted(u for x in a for u in x) == [(w, h) for w in range(W) for h in range(halves)]
                assert max(len(x) for x in a) - min(len(x) for x in a) <= 1
                sch = D.UnitSchedule(W, R, 1, halves)
                for r in range(R):                      # slots follow the call order (what pack_rows relies on)
                    rows = [(w, h) for w, hs in sch.calls(r) for h in hs]
                    assert [sch.slot[u] for u in rows] == [(r, i) for i in range(len(rows))]
    a = D.partition_units(10, 8)
    assert [len(x) for x in a] == [3, 3, 3, 3, 2, 2, 2, 2]
    for r in range(4):
        lone = [u for u in a[r] if sum(1 for v in a[r] if v[0] == u[0]) == 1]
        assert len(lone) == 1 and lone[0][1] == 0, a[r]          # the lone half of a heavy rank is the uncond one
    assert a[4] == [(4, 1), (5, 1)] and a[5] == [(6, 1), (7, 1)] and a[6] == [(8, 0), (8, 1)]
    assert D.partition_units(10, 4) == [[(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)], [(2, 1), (3, 0), (3, 1), (4, 0), (4, 1)],
                                        [(5, 0), (5, 1), (6, 0), (6, 1), (7, 0)], [(7, 1), (8, 0), (8, 1), (9, 0), (9, 1)]]


def test_frame_shard_schedule_and_auto_policy():
    D = distributed
    # a 16-frame clip (1 window = 2 units) on 8 ranks: 4 ranks per unit; long clips keep S = 1
    assert D.choose_frame_shards(1, 8, 16, 64) == 4 and D.choose_frame_shards(10, 8, 16, 64) == 1
    assert D.choose_frame_shards(1, 2, 16, 64) == 1 and D.choose_frame_shards(2, 8, 16, 64) == 2
    assert D.choose_frame_shards(1, 8, 12, 64) == 4 and D.choose_frame_shards(1, 8, 6, 64) == 2    # S | window length
    assert D.choose_frame_shards(1, 8, 16, 2) == 2                                                   # S | coarsest hw
    u = D.UnitSchedule(1, 8, 4)
    assert u.groups == 2 and u.calls(0) == u.calls(3) == [(0, [0])] and u.calls(4) == u.calls(7) == [(0, [1])]
    assert u.unit_ranks((0, 0)) == ([0, 1, 2, 3], 0) and u.unit_ranks((0, 1)) == ([4, 5, 6, 7], 0)
    u = D.UnitSchedule(3, 4, 2)                              # 6 units over 2 groups of 2 ranks
    assert [u.calls(r) for r in range(4)] == [[(0, [0, 1]), (1, [0])]] * 2 + [[(1, [1]), (2, [0, 1])]] * 2
    with pytest.raises(ValueError):
        D.UnitSchedule(1, 6, 4)


def test_mixed_schedule_whole_units_plus_sharded_leftovers():
    """distributed.MixedUnitSchedule (round 3): when units % world != 0 and the left-over units fill the node exactly
    once as S-way frame-sharded units, every rank carries floor(units / world) whole units + 1/S of one more.  Checks
    the policy, that every frame granule of every unit has exactly one (rank, slot), that a rank's slots are the
    consecutive ones pack_rows relies on, and the per-rank load (2.5 unit-times for the config-4 clip on 8 GPUs against
    3 for any whole-unit schedule)."""
    D = distributed
    assert D.choose_mixed_shards(20, 8, 16, 64) == 2          # BASELINE configs[3]: 10 windows x 2 halves on 8 GPUs
    assert D.choose_mixed_shards(10, 4, 8, 4) == 2 and D.choose_mixed_shards(10, 8, 8, 4) == 4
    assert D.choose_mixed_shards(20, 4, 16, 64) == 1          # divides evenly: nothing to shard
    assert D.choose_mixed_shards(2, 8, 16, 64) == 1           # fewer units than ranks: choose_frame_shards' case
    assert D.choose_mixed_shards(22, 8, 16, 64) == 1          # 6 left-over units do not tile 8 ranks
    assert D.choose_mixed_shards(20, 8, 15, 64) == 1 and D.choose_mixed_shards(20, 8, 16, 1) == 1   # S | f, S | hw
    for W, R, S in ((10, 8, 2), (5, 4, 2), (5, 8, 4), (3, 4, 2)):
        m = D.MixedUnitSchedule(W, R, S)
        units = [(w, h) for w in range(W) for h in range(2)]
        assert sorted(m.slots) == units and all(len(v) == S for v in m.slots.values())
        seen = set()
        for u, gl in m.slots.items():
            for (r, s) in gl:
                assert 0 <= r < R and 0 <= s < m.max_slots and (r, s) not in seen
                seen.add((r, s))
        assert len(seen) == R * m.max_slots                   # every slot of every rank is used: equal load
        for r in range(R):
            rows = [(w, h) for w, hs in m.whole_calls(r) for h in hs]
            assert [m.slots[u] for u in rows] == [[(r, k * S + j) for j in range(S)] for k in range(len(rows))]
            su = m.split_unit(r)
            assert m.slots[su][r % S] == (r, len(rows) * S)   # this rank's granule of the shared unit: its last slot
        assert m.rounds() == len(units) // R + 1.0 / S
    assert D.MixedUnitSchedule(10, 8, 2).rounds() == 2.5
    with pytest.raises(ValueError):
        D.MixedUnitSchedule(10, 8, 4)


def test_subgroup_creation_order_is_identical_on_every_rank_at_world_8(monkeypatch):
    """`dist.new_group` is collective over the WHOLE world and must be called by every rank in the same order with the
    same rank lists (the classic RCCL deadlock when a rank skips or reorders one).  Replays, for each of the 8 ranks of
    the config-4 clip's mixed schedule (and the uniform frame-shard schedules), exactly the `DistContext.frame_shard`
    calls `VExpressPipeline.denoise` makes, with `dist.new_group` recorded instead of executed."""
    D = distributed
    world = 8

    def calls_of(rank, shard_requests):
        seq = []
        monkeypatch.setattr(dist, "new_group", lambda ranks: (seq.append(tuple(ranks)), ("grp", tuple(ranks)))[1])
        dc = D.DistContext(rank, world, None)
        mine = []
        for S in shard_requests:
            fs = dc.frame_shard(S)
            mine.append(None if fs is None else (fs.index, fs.size, fs.group))
        return seq, mine

    # mixed schedule of pipeline.denoise: plan_calls = [(whole units, 1), (the shared unit, Sm)] -> frame_shard(1), frame_shard(Sm)
    for requests in ([1, 2], [1, 4], [2], [4], [8], [2, 2, 4, 2]):
        seqs = [calls_of(r, requests) for r in range(world)]
        first = seqs[0][0]
        for r, (seq, mine) in enumerate(seqs):
            assert seq == first, f"rank {r} creates its sub-groups in a different order: {seq} vs {first}"
            for S, got in zip(requests, mine):
                if S == 1:
                    assert got is None
                else:
                    g0 = (r // S) * S
                    assert got == (r % S, S, ("grp", tuple(range(g0, g0 + S)))), (r, S, got)
        # every group of every size is created exactly once, in ascending order of its first rank
        want = []
        for S in dict.fromkeys(s for s in requests if s > 1):
            want += [tuple(range(g * S, (g + 1) * S)) for g in range(world // S)]
        assert list(first) == want


def _worker_groups8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dc = distributed.DistContext.from_env()
    got = []
    with distributed.CommTimer() as ct:
        for S in (1, 2, 4):                                   # the mixed schedule asks for 1 then Sm; 4 = the 5-window clip
            fs = dc.frame_shard(S)
            if fs is None:
                continue
            b, f_loc, hw, c = 2, 8 // S, 8, 3
            x = torch.arange(b * f_loc * hw * c, dtype=torch.float32).view(b * f_loc, hw, c) + 1000.0 * rank
            px = fs.to_pixel_shard(x, b, f_loc)
            back = fs.to_frame_shard(px, b, f_loc)
            got.append((S, torch.equal(back, x), tuple(px.shape)))
        gathered = dc.all_gather_units(torch.full((2, 4), float(rank)), 2)
    q.put((rank, got, gathered[:, 0, 0].tolist(), {k: v["calls"] for k, v in ct.summary().items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_world_8_sub_groups_and_exchanges_over_gloo():
    """8 real processes: sub-groups of 2 and 4 consecutive ranks created through DistContext.frame_shard in the order the
    pipeline uses, a layout round trip through each group's all-to-all, the world-wide all-gather, and CommTimer's
    bookkeeping (what bench.py --gpus N reports as the collectives' share of a step)."""
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_groups8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, *_ in results) == list(range(world))
    for rank, got, owners, calls in results:
        assert got == [(2, True, (2 * 8, 4, 3)), (4, True, (2 * 8, 2, 3))], (rank, got)
        assert owners == [float(r) for r in range(world)]
        assert calls == {"all_to_all": 4, "all_gather": 1}, calls
