"""First contact of the collective code path with RCCL on the hardware that exists: ONE GPU, a one-rank "nccl" group.

No multi-GPU box was available in any round, so `distributed.py`'s RCCL branch (`dist.get_backend() == "nccl"`: device
tensors handed to the collective as they are, no host staging) had never executed.  A one-rank group cannot measure
xGMI and is NOT a scaling number, but it does run what breaks first when such a design meets RCCL: the eager
`init_process_group("nccl", device_id=...)`, `new_group` on RCCL, `all_gather_into_tensor` / `all_to_all_single` /
`broadcast` / `barrier` on the views and dtypes the data path hands them, and the sharded denoising loop + frame-split
decode issuing them between kernels on the same stream order.  Checks:
  1. the raw collectives on the data path's own buffers (values unchanged at world size 1),
  2. FrameShard's two layout switches are inverses of each other through the RCCL all-to-all,
  3. `VExpressPipeline.denoise` + `decode_latents` through the collective path == the sequential path, bit for bit.
    python tools/rccl_world1_probe.py [ddim steps = 3] > gpurun_out/<tag>_rccl_world1.json"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v_express_amd as vx  # noqa: E402
from v_express_amd import ops, synth  # noqa: E402
from v_express_amd.context import uniform  # noqa: E402
from v_express_amd.distributed import CommTimer, DistContext, FrameShard  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    res = {"world_size": 1, "measurement": False,
           "note": "one-rank RCCL group on one GPU: executes the RCCL branch of distributed.py, measures no link"}
    t0 = time.time()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=dev)
    res["init_s"] = round(time.time() - t0, 3)
    res["backend"] = dist.get_backend()
    dc = DistContext(0, 1, None, force=True)
    assert dc.enabled and dc.backend == "nccl"

    # 1. raw collectives on the buffers of configs[1]
    checks = {}
    with CommTimer() as tm:
        dist.barrier()
        lat = torch.randn(1, 4, 16, 64, 64, device=dev)
        want = lat.clone()
        dc.broadcast(lat, src=0)
        checks["broadcast_keeps_values"] = bool(torch.equal(lat, want))
        local = torch.randn(2, 16 * 4096, 4, device=dev)                # one timestep's exchange buffer (2 units)
        g = dc.all_gather_units(local, 2)
        checks["all_gather_units"] = bool(g.shape == (1, 2, 16 * 4096, 4) and torch.equal(g[0], local))
        frames = torch.randn(16, 3, 512, 512, device=dev)
        g = dc.all_gather_frames(frames)
        checks["all_gather_frames"] = bool(torch.equal(g[0], frames))
        # 2. the frame-shard layout switches over a sub-group created on RCCL
        grp = dist.new_group([0])
        fs = FrameShard(0, 1, grp)
        x = torch.randn(2 * 16, 4096, 320, device=dev).to(torch.bfloat16)          # [b * f, hw, C] of the 64x64 level
        px = fs.to_pixel_shard(x, 2, 16)
        back = fs.to_frame_shard(px, 2, 16)
        checks["frame_shard_round_trip"] = bool(torch.equal(back, x) and torch.equal(px, x))
        torch.cuda.synchronize()
    res["raw_collectives"] = tm.summary()
    res["checks"] = checks

    # 3. the denoising loop + decode through the collective path against the sequential path
    cfg, vcfg = synth.UNetConfig(), synth.VaeConfig()
    F, ctx, ovl, h, w = 28, 16, 4, 64, 64                    # two overlapping windows -> 4 units, one all-gather per step
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
    timesteps = sched.timesteps.tolist()[:steps]
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

    lat_seq, vid_seq = clip(DistContext())
    with CommTimer() as tm:
        lat_rccl, vid_rccl = clip(dc)
        sched_used = dict(pipe.last_schedule)
    res["loop"] = {"frames": F, "windows": len(windows), "ddim_steps": steps, "schedule": sched_used,
                   "latents_bit_identical": bool(torch.equal(lat_seq, lat_rccl)),
                   "video_bit_identical": bool(torch.equal(vid_seq, vid_rccl)),
                   "finite": bool(torch.isfinite(vid_rccl).all()),
                   "collectives": tm.summary()}
    dist.barrier()
    dist.destroy_process_group()
    ok = all(checks.values()) and res["loop"]["latents_bit_identical"] and res["loop"]["video_bit_identical"] and res["loop"]["finite"]
    res["ok"] = bool(ok)
    print(json.dumps(res, indent=1))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
