"""Multi-GPU pre-flight: N ranks, one per GPU, over RCCL - run BEFORE any scaling bench on the day a
multi-GPU node appears, so that first contact produces a diagnosis instead of a hang.

    python tools/rccl_probe.py [--gpus 2] [--steps 2]          (re-launches itself under torch.distributed.run)
    timeout 60 python tools/rccl_probe.py --gpus 2             (tools/gpu_job.sh scale8 does this first)

Every step announces itself on stderr BEFORE it starts ("[rank r] step: ...") and reports its wall time after, so the
last line of a hung or killed run names the call that did not return.  Steps (each a collective on the data path's own
buffers, shapes and dtypes):
  init          init_process_group("nccl", device_id=cuda:LOCAL_RANK)     (eager communicator creation)
  barrier       dist.barrier()
  broadcast     rank 0's fp32 latents [1, 4, 16, 64, 64] to every rank, checked
  all_gather    all_gather_into_tensor of one timestep's exchange buffer (distributed.DistContext.all_gather_units), checked
  subgroups     new_group for every pair of consecutive ranks (what the frame-sharded / mixed schedules create), then
                all_to_all_single on the pair: FrameShard's two layout switches are inverses of each other, checked
  loop          VExpressPipeline.denoise (28 frames = two overlapping windows = 4 units, `--steps` DDIM steps) +
                decode_latents through the sharded path; rank 0 then runs the SAME clip sequentially and compares bit for bit
Rank 0 prints one JSON object (ok, per-step seconds, per-collective ms of the loop).  A one-GPU box cannot run it (RCCL refuses
two ranks on one device: profiles/r05z_rccl_two_ranks_one_gpu.txt); the one-rank twin is tools/rccl_world1_probe.py."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(json.dumps({"ok": False, "skipped": True, "reason": f"{have} GPU(s) visible, the probe needs {args.gpus}"}))
            sys.exit(3)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        raise SystemExit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                          f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port",
                                          str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    times, checks = {}, {}

    class step:
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            print(f"[rank {rank}] step: {self.name} ...", file=sys.stderr, flush=True)
            self.t0 = time.time()

        def __exit__(self, et, ev, tb):
            torch.cuda.synchronize()
            times[self.name] = round(time.time() - self.t0, 3)
            print(f"[rank {rank}] step: {self.name} {'FAILED' if et else 'done'} in {times[self.name]} s", file=sys.stderr, flush=True)

    with step("init"):
        dist.init_process_group("nccl", device_id=dev)
    import v_express_amd as vx
    from v_express_amd import ops, synth
    from v_express_amd.context import uniform
    from v_express_amd.distributed import CommTimer, DistContext, FrameShard
    dc = DistContext.from_env()
    assert dc.enabled and dc.backend == "nccl" and dc.world_size == world
    with step("barrier"):
        dist.barrier()
    with step("broadcast"):
        g = torch.Generator(device=dev).manual_seed(7)                        # same seed: every rank knows rank 0's values
        want = torch.randn(1, 4, 16, 64, 64, device=dev, generator=g)
        lat = want.clone() if rank == 0 else torch.zeros_like(want)
        dc.broadcast(lat, src=0)
        checks["broadcast"] = bool(torch.equal(lat, want))
    with step("all_gather"):
        local_units = torch.full((2, 16 * 4096, 4), float(rank + 1), device=dev)
        got = dc.all_gather_units(local_units, 2)
        checks["all_gather"] = bool(got.shape == (world, 2, 16 * 4096, 4) and
                                    all(bool((got[r] == r + 1).all()) for r in range(world)))
    with step("subgroups"):
        if world % 2 == 0:
            fs = dc.frame_shard(2)                                             # collective: new_group for every pair
            gx = torch.Generator(device=dev).manual_seed(100 + rank)
            x = torch.randn(2 * 8, 4096, 320, device=dev, generator=gx).to(torch.bfloat16)   # this rank's 8 of 16 frames
            px = fs.to_pixel_shard(x, 2, 8)
            back = fs.to_frame_shard(px, 2, 8)
            checks["frame_shard_round_trip"] = bool(px.shape == (32, 2048, 320) and torch.equal(back, x))
        else:
            checks["frame_shard_round_trip"] = None
    with step("model"):
        cfg, vcfg = synth.UNetConfig(), synth.VaeConfig()
        F, ctx, ovl, h, w = 28, 16, 4, 64, 64
        unet = vx.UNet3DConditionModel(cfg).to(dev)
        refnet = vx.UNet2DConditionModel(cfg).to(dev)
        vae = vx.AutoencoderKLDecoder(vcfg).to(dev)
        unet.load_state_dict(synth.unet3d_state_dict(cfg, seed=42, device=dev, dtype=torch.bfloat16, draw_on_device=True))
        unet.release_raw_weights()
        refnet.load_state_dict(synth.refnet_state_dict(cfg, seed=43, device=dev, dtype=torch.bfloat16, draw_on_device=True))
        refnet.release_raw_weights()
        vae.load_state_dict(synth.vae_decoder_state_dict(vcfg, seed=44, device=dev, dtype=torch.bfloat16, draw_on_device=True))
        vae._prepared()
        sched = vx.DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                                 steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True,
                                 timestep_spacing="trailing")
        pipe = vx.VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet, scheduler=sched)
        inp = synth.synthetic_inputs(cfg, F, h, w, seed=42, device=dev)
        writer = vx.ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
        reader = vx.ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                              reference_attention_weight=0.95, audio_attention_weight=3.0)
        refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768, device=dev), return_dict=False)
        reader.update(writer, True)
        sched.set_timesteps(25)
        timesteps = sched.timesteps.tolist()[:args.steps]
        windows = list(uniform(step=0, num_frames=F, context_size=ctx, context_stride=1, context_overlap=ovl, closed_loop=False))
        c0 = cfg.block_out_channels[0]
        kps_tokens = ops.ncfhw_to_nhwc(inp["kps_features"], c0).view(2, F, h * w, c0)
        audio = inp["audio_embeddings"].to(torch.bfloat16).contiguous()

    def clip(ctx_obj):
        pipe.dist = ctx_obj
        lat = inp["latents"].clone()
        pipe.denoise(lat, kps_tokens, audio, timesteps, windows, 3.5)
        video = pipe.decode_latents(lat)
        torch.cuda.synchronize()
        return lat, video

    with step("loop"):
        with CommTimer() as tm:
            lat_d, vid_d = clip(dc)
            schedule = dict(pipe.last_schedule)
        coll = tm.summary()
    loop = {"frames": F, "windows": len(windows), "ddim_steps": args.steps, "schedule": schedule, "collectives": coll}
    if rank == 0:
        with step("sequential_reference"):
            lat_s, vid_s = clip(DistContext())
        loop["latents_bit_identical"] = bool(torch.equal(lat_s, lat_d))
        loop["video_bit_identical"] = bool(torch.equal(vid_s, vid_d))
        loop["finite"] = bool(torch.isfinite(vid_d).all())
    with step("final_barrier"):
        dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        ok = (all(v is not False for v in checks.values()) and loop["latents_bit_identical"] and loop["video_bit_identical"]
              and loop["finite"])
        print(json.dumps({"ok": bool(ok), "world_size": world, "measurement": False, "backend": "nccl", "checks": checks,
                          "step_seconds": times, "loop": loop,
                          "note": "pre-flight: correctness of every collective of the data path on real links; not a scaling number"},
                         indent=1))
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
