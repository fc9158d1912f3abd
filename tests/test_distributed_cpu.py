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


def run_loop(rank, world, F, cs, co, steps, dc):
    windows = OL.uniform_windows(F, cs, co)
    plan = context.overlap_plan(windows, F)
    f = len(windows[0])
    sch = distributed.UnitSchedule(len(windows), world)
    s = DDIMScheduler(**SCHED_KW)
    s.set_timesteps(steps)
    lat = torch.randn(1, 4, F, 4, 4, generator=torch.Generator().manual_seed(0))
    for t in s.timesteps.tolist():
        local = torch.zeros(sch.max_units, 4, f, 4, 4)
        for w, halves in sch.calls(rank):
            for hlf in halves:
                local[sch.slot[(w, hlf)][1]] = fake_unit(lat, windows[w], hlf, t)
        gathered = dc.all_gather_units(local, sch.max_units)
        sa, s1a, sap, s1ap = s.step_coefficients(t)
        new = lat.clone()
        for fr in plan["step_frames"]:
            v = None
            for (wi, li) in plan["terms"][fr]:
                u = gathered[sch.slot[(wi, 0)]][:, li]
                c = gathered[sch.slot[(wi, 1)]][:, li]
                term = (u + 3.5 * (c - u)) / float(plan["counts"][fr])
                v = term if v is None else v + term
            x = lat[0, :, fr]
            new[0, :, fr] = sap * (sa * x - s1a * v) + s1ap * (sa * v + s1a * x)
        lat = new
    return lat


def _worker(rank, world, port, F, cs, co, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dc = distributed.DistContext.from_env()
    assert dc.enabled and dc.world_size == world and dc.rank == rank
    out = run_loop(rank, world, F, cs, co, steps, dc)
    lo, hi = distributed.split_frames(F, world)[rank]
    frames = torch.zeros(distributed.split_frames(F, world)[0][1], 3)
    frames[:hi - lo] = float(rank + 1)
    allf = dc.all_gather_frames(frames).reshape(-1, 3)[:F]
    q.put((rank, out, allf[:, 0].clone()))
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
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, owner in results:
        assert torch.equal(out, ref), f"rank {rank} diverged from the single-process loop"
        lo, hi = distributed.split_frames(F, world)[0]
        assert (owner[:hi] == 1).all() and (owner[hi:] == 2).all()
