"""Where one DDIM step goes, by block type and UNet level (GPU box): HIP events around every resnet block, spatial transformer
(with its four parts: proj_in, self-attention, reference attention, audio cross-attention, feed-forward + proj_out), motion
module and resampler of one CFG forward at BASELINE configs[1].  python tools/step_breakdown.py > gpurun_out/<tag>_step_breakdown.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
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

    recs = []

    def timed(name, fn, level_of):
        def run(*a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            recs.append((name, level_of(a, k), s, e))
            return r
        return run

    lvl_hw = lambda a, k: k.get("H") or a[3]
    B.resnet_block = timed("resnet", B.resnet_block, lambda a, k: a[3])
    B.spatial_transformer_read = timed("spatial_transformer", B.spatial_transformer_read, lambda a, k: k["H"])
    B.motion_module = timed("motion_module", B.motion_module, lambda a, k: k["H"])
    B.downsample = timed("downsample", B.downsample, lambda a, k: a[3])
    B.upsample = timed("upsample", B.upsample, lambda a, k: a[3])
    # parts of the spatial transformer
    B._norm_proj_in = timed("  st.norm_proj_in", B._norm_proj_in, lambda a, k: a[3])
    B._self_attention = timed("  st.self_attention", B._self_attention, lambda a, k: k["n_tok"])
    B._feed_forward = timed("  st/mm.feed_forward", B._feed_forward, lambda a, k: a[1].shape[0])
    _ax, _att = ops.audio_xattn, ops.attention
    ops.audio_xattn = timed("  st.audio_xattn", _ax, lambda a, k: k["rows_per_frame"])
    ops.attention = timed("  attention(kernel)", _att, lambda a, k: (k["n_q"], k["n_kv"]))

    for _ in range(2):
        pipe.denoise(inp["latents"].clone(), kps, audio, ts[:1], windows, 3.5)
    torch.cuda.synchronize()
    recs.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    pipe.denoise(inp["latents"].clone(), kps, audio, ts[:1], windows, 3.5)
    e1.record()
    torch.cuda.synchronize()
    total = e0.elapsed_time(e1)
    print(f"# one DDIM step with block-level events: {total:.2f} ms")
    by = {}
    for name, lvl, s, e in recs:
        d = by.setdefault((name, str(lvl)), [0, 0.0])
        d[0] += 1
        d[1] += s.elapsed_time(e)
    tot = {}
    for (name, lvl), (n, ms) in sorted(by.items(), key=lambda kv: (kv[0][0].strip(), -kv[1][1])):
        print(f"{name:24s} level {lvl:>14s} x{n:3d} {ms:8.3f} ms  {100 * ms / total:5.1f} %  {1e3 * ms / n:8.1f} us each")
        t = tot.setdefault(name, 0.0)
        tot[name] = t + ms
    print("# per block type")
    for name, ms in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"{name:24s} {ms:8.3f} ms  {100 * ms / total:5.1f} %")


if __name__ == "__main__":
    main()
