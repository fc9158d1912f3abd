"""Kernel-by-kernel timeline of ONE block call inside a DDIM step (GPU box): torch.profiler (roctracer) around a step, the
launches between the start and end of the n-th call of a blocks.* function at a given level, in start order with durations and
the gaps between them.   python tools/block_trace.py spatial_transformer_read 16 [call index]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    fn_name = sys.argv[1] if len(sys.argv) > 1 else "spatial_transformer_read"
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    which = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    import v_express_amd as vx
    from v_express_amd import blocks as B, ops, synth
    from v_express_amd.context import uniform
    dev = torch.device("cuda", 0)
    elem = torch.bfloat16
    cfg = synth.UNetConfig()
    F, h = 16, 64
    unet = vx.UNet3DConditionModel(cfg).to(dev).to(elem)
    refnet = vx.UNet2DConditionModel(cfg).to(dev).to(elem)
    unet.load_state_dict(synth.unet3d_state_dict(cfg, seed=42, device=dev, dtype=elem, draw_on_device=True))
    unet.release_raw_weights()
    refnet.load_state_dict(synth.refnet_state_dict(cfg, seed=43, device=dev, dtype=elem, draw_on_device=True))
    refnet.release_raw_weights()
    sched = vx.DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                             steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True,
                             timestep_spacing="trailing")
    vae = vx.AutoencoderKLDecoder(synth.VaeConfig()).to(dev).to(elem)
    pipe = vx.VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet, scheduler=sched)
    inp = synth.synthetic_inputs(cfg, F, h, h, seed=42, device=dev)
    writer = vx.ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    reader = vx.ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                          reference_attention_weight=0.95, audio_attention_weight=3.0)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768, device=dev), return_dict=False)
    reader.update(writer, True)
    sched.set_timesteps(25)
    ts = sched.timesteps.tolist()
    windows = list(uniform(step=0, num_frames=F, context_size=16, context_stride=1, context_overlap=4, closed_loop=False))
    c0 = cfg.block_out_channels[0]
    kps = ops.ncfhw_to_nhwc(inp["kps_features"], c0).view(2, F, h * h, c0)
    audio = inp["audio_embeddings"].to(elem).contiguous()
    for _ in range(2):
        pipe.denoise(inp["latents"].clone(), kps, audio, ts[:1], windows, 3.5)
    torch.cuda.synchronize()

    orig = getattr(B, fn_name, None)
    count = [0]

    marker = torch.zeros(64, device=dev)

    def wrapped(*a, **k):
        lvl = k.get("H") if "H" in k else a[3]
        if lvl == level:
            idx = count[0]
            count[0] += 1
            if idx == which:
                marker.cos_()              # (a kernel name nothing else in the step launches: the range markers)
                r = orig(*a, **k)
                marker.cos_()
                return r
        return orig(*a, **k)
    whole = fn_name in ("step", "vae")
    if not whole:
        setattr(B, fn_name, wrapped)
    if fn_name == "vae":
        vae.load_state_dict(synth.vae_decoder_state_dict(synth.VaeConfig(), seed=44, device=dev, dtype=elem, draw_on_device=True))
        vae._prepared()
        vae.decode_video(inp["latents"][:, :, :8].contiguous())
        torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        if fn_name == "vae":
            marker.cos_()
            vae.decode_video(inp["latents"][:, :, :8].contiguous())     # one chunk of 8 frames (a 16-frame clip = two)
            marker.cos_()
        elif whole:
            # two DDIM steps, markers around the SECOND (the first carries the once-per-clip work: audio K | V, operand packs)
            lat = inp["latents"].clone()
            orig_step = ops.overlap_ddim_step
            n_calls = [0]

            def step_marked(*a, **k):
                r = orig_step(*a, **k)
                n_calls[0] += 1
                marker.cos_()
                return r
            ops.overlap_ddim_step = step_marked
            pipe.denoise(lat, kps, audio, ts[:2], windows, 3.5)
            ops.overlap_ddim_step = orig_step
        else:
            pipe.denoise(inp["latents"].clone(), kps, audio, ts[:1], windows, 3.5)
        torch.cuda.synchronize()
    dev_evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA),
                     key=lambda e: e.time_range.start)
    marks = [i for i, e in enumerate(dev_evs) if "cos" in e.name.lower()]
    if len(marks) < 2:
        print("markers not found among", len(dev_evs), "device events")
        return
    kern = dev_evs[marks[0] + 1:marks[1]]
    print(f"# {fn_name} level {level} call {which}: {len(kern)} launches")
    if whole:
        by = {}
        for e in kern:
            d = by.setdefault(e.name[:100], [0, 0.0])
            d[0] += 1
            d[1] += e.time_range.end - e.time_range.start
        tot = sum(v[1] for v in by.values())
        span = kern[-1].time_range.end - kern[0].time_range.start
        print(f"# kernel time {tot / 1e3:.2f} ms, span {span / 1e3:.2f} ms, idle {(span - tot) / 1e3:.2f} ms")
        for name, (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            print(f"{us / 1e3:8.3f} ms {100 * us / tot:5.1f} % x{n:4d} {us / n:8.1f} us  {name}")
        return
    tot, prev_end = 0.0, None
    for e in kern:
        st, du = e.time_range.start, e.time_range.end - e.time_range.start
        gap = 0.0 if prev_end is None else st - prev_end
        prev_end = st + du
        tot += du
        print(f"{du:9.1f} us  gap {gap:7.1f}  {e.name[:120]}")
    print(f"# sum of durations {tot:.1f} us, span {kern[-1].time_range.end - kern[0].time_range.start:.1f} us")


if __name__ == "__main__":
    main()
