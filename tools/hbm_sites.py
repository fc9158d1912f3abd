"""Per-site table of the HBM-bound launches of ONE DDIM step at BASELINE configs[1] (512x512, 16 frames, CFG): every
`ops._hbm_op` record grouped by (wrapper name, algorithmic bytes) -> launches, HIP-event time, GB/s.  Answers "which
GroupNorm sites still run their own statistics pass, and what does each cost" (bench.py's hbm_kernels is the per-name sum).
GPU box only:  python tools/hbm_sites.py [--size 512] > gpurun_out/<tag>_hbm_sites.txt"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frames", type=int, default=16)
    args = ap.parse_args()
    import v_express_amd as vx
    from v_express_amd import ops, synth
    from v_express_amd.context import uniform
    dev = torch.device("cuda", 0)
    elem = torch.bfloat16
    cfg = synth.UNetConfig()
    F, h = args.frames, args.size // 8
    unet = vx.UNet3DConditionModel(cfg).to(dev).to(elem)
    refnet = vx.UNet2DConditionModel(cfg).to(dev).to(elem)
    unet.load_state_dict(synth.unet3d_state_dict(cfg, seed=42, device=dev, dtype=elem, draw_on_device=True))
    unet.release_raw_weights()
    refnet.load_state_dict(synth.refnet_state_dict(cfg, seed=43, device=dev, dtype=elem, draw_on_device=True))
    refnet.release_raw_weights()
    sched = vx.DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                             steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True,
                             timestep_spacing="trailing")
    vcfg = synth.VaeConfig()
    vae = vx.AutoencoderKLDecoder(vcfg).to(dev).to(elem)
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
    prof = ops.OpProfile()
    gprof = ops.GemmProfile()
    with prof, gprof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pipe.denoise(inp["latents"].clone(), kps, audio, ts[:1], windows, 3.5)
        e1.record()
    torch.cuda.synchronize()
    step_ms = e0.elapsed_time(e1)
    by = {}
    for name, nbytes, s, e, fl, sym, _ in prof.records:
        d = by.setdefault((name, int(nbytes)), [0, 0.0])
        d[0] += 1
        d[1] += s.elapsed_time(e) * 1e-3
    print(f"# one instrumented DDIM step (events around every launch): {step_ms:.2f} ms")
    tot = {}
    for (name, nbytes), (n, sec) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:26s} {nbytes / 1e6:9.2f} MB x{n:3d}  {1e6 * sec / n:8.1f} us  {1e3 * sec:7.3f} ms  {nbytes * n / sec / 1e9:7.0f} GB/s")
        t = tot.setdefault(name, [0, 0.0, 0.0])
        t[0] += n; t[1] += sec; t[2] += nbytes * n
    print("# per name")
    for name, (n, sec, b) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:26s} x{n:3d} {1e3 * sec:7.3f} ms  {b / sec / 1e9:7.0f} GB/s")
    gs = sum(r[0].elapsed_time(r[1]) for r in gprof.records)
    print(f"# vx_gemm launches: {len(gprof.records)}  {gs:.2f} ms")


if __name__ == "__main__":
    main()
