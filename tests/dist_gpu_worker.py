"""torch.distributed.run worker for tests/test_gpu_models.py: the small-config denoising loop with the ranks folded onto
ONE GPU (VX_DIST_BACKEND=gloo: collectives staged through the host) - exercises the real process groups, the
frame-shard all-to-alls inside the motion modules and the unit gather.  Rank 0 saves the final latents.
usage: dist_gpu_worker.py OUT.pt F CONTEXT OVERLAP STEPS FRAME_SHARDS(0 = automatic)"""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import cases  # noqa: E402


def emulate_kernels():
    """CPU suite only: tests/fake_ops.py stands in for the HIP wrappers and the device guards are lifted, so the whole
    host side (models, loop, sharding) runs in a process without a GPU."""
    import fake_ops
    from v_express_amd import ops, unet_3d, vae

    class _Patch:
        def setattr(self, obj, name, value):
            setattr(obj, name, value)
    fake_ops.install(_Patch(), ops)
    unet_3d._UNetBase._need_gpu = lambda self: None
    vae.AutoencoderKLDecoder._need_gpu = lambda self: None
    if os.environ.get("VX_TEST_FORCE_ROUND4") == "1":
        force_round4_paths(ops)


def force_round4_paths(ops, patch=setattr):
    """The round-4 host paths at the small widths of the CPU models, where the routing rules would not pick them: every
    temporal attention block as ONE `ops.tblock_fused` call (also in the pixel-shard layout of a frame-sharded unit), row
    statistics as two-part sums ([rows, 4] buffers) at every width."""
    patch(ops, "tblock_fused_applies", lambda c, heads, f, hw: True)
    patch(ops, "STATS_PARTS_WIDTHS", set(range(8, 4096, 8)))


def build_pipeline(device):
    """The small-config pipeline (UNet3D + ReferenceNet + VAE decoder, seeded synthetic weights)."""
    from v_express_amd import (AutoencoderKLDecoder, DDIMScheduler, UNet2DConditionModel, UNet3DConditionModel,
                               VExpressPipeline, synth)
    cfg = cases.unet_cfg(cases.SMALL)
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    unet = UNet3DConditionModel(cfg).to(device)
    refnet = UNet2DConditionModel(cfg).to(device)
    unet.load_state_dict(synth.unet3d_state_dict(cfg), strict=True)
    refnet.load_state_dict(synth.refnet_state_dict(cfg), strict=True)
    vae = AutoencoderKLDecoder(vcfg).to(device)
    vae.load_state_dict(synth.vae_decoder_state_dict(vcfg))
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                          steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True,
                          timestep_spacing="trailing")
    return VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet, scheduler=sched)


def run(F, cf, co, steps, frame_shards, latent=16, device="cuda"):
    from v_express_amd import ReferenceAttentionControl, ops, synth
    cfg = cases.unet_cfg(cases.SMALL)
    pipe = build_pipeline(device)
    unet, refnet, sched = pipe.denoising_unet, pipe.reference_net, pipe.scheduler
    return _run(pipe, unet, refnet, sched, cfg, F, cf, co, steps, frame_shards, latent, device)


def _run(pipe, unet, refnet, sched, cfg, F, cf, co, steps, frame_shards, latent, device):
    from v_express_amd import ReferenceAttentionControl, ops, synth
    pipe.frame_shards = frame_shards or None
    inp = synth.synthetic_inputs(cfg, F, latent, latent)
    if device != "cuda":
        # the pieces of VExpressPipeline.__call__ (which times itself with CUDA events), in its order
        from v_express_amd.context import get_context_scheduler
        writer = ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
        reader = ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                           reference_attention_weight=cases.W_REF,
                                           audio_attention_weight=cases.W_AUD)
        refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768), return_dict=False)
        reader.update(writer, True)
        sched.set_timesteps(steps)
        windows = list(get_context_scheduler("uniform")(step=0, num_frames=F, context_size=cf, context_stride=1,
                                                        context_overlap=co, closed_loop=False))
        c0 = cfg.block_out_channels[0]
        kps = ops.ncfhw_to_nhwc(inp["kps_features"], c0).view(2, F, latent * latent, c0)
        audio = inp["audio_embeddings"].to(torch.bfloat16).contiguous()
        lat = inp["latents"].clone().float()
        pipe.denoise(lat, kps, audio, sched.timesteps.tolist(), windows, cases.GUIDANCE)
        return lat
    lat = pipe(None, None, None, latent * 8, latent * 8, F, steps, cases.GUIDANCE, context_frames=cf,
               context_overlap=co, reference_attention_weight=cases.W_REF, audio_attention_weight=cases.W_AUD,
               reference_latents=inp["ref_latents"], kps_features=inp["kps_features"],
               audio_embeddings=inp["audio_embeddings"], latents=inp["latents"], decode=False)
    return lat.detach().cpu()


def main():
    out, F, cf, co, steps, S = sys.argv[1], *map(int, sys.argv[2:7])
    dist.init_process_group(os.environ.get("VX_DIST_BACKEND", "gloo"))
    torch.cuda.set_device(0)
    lat = run(F, cf, co, steps, S)
    if dist.get_rank() == 0:
        torch.save(lat, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
