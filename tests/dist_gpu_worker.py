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


def run(F, cf, co, steps, frame_shards, latent=16):
    from v_express_amd import (AutoencoderKLDecoder, DDIMScheduler, UNet2DConditionModel, UNet3DConditionModel,
                               VExpressPipeline, synth)
    cfg = cases.unet_cfg(cases.SMALL)
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    unet = UNet3DConditionModel(cfg).to("cuda")
    refnet = UNet2DConditionModel(cfg).to("cuda")
    unet.load_state_dict(synth.unet3d_state_dict(cfg), strict=True)
    refnet.load_state_dict(synth.refnet_state_dict(cfg), strict=True)
    vae = AutoencoderKLDecoder(vcfg).to("cuda")
    vae.load_state_dict(synth.vae_decoder_state_dict(vcfg))
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                          steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True,
                          timestep_spacing="trailing")
    pipe = VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet, scheduler=sched)
    pipe.frame_shards = frame_shards or None
    inp = synth.synthetic_inputs(cfg, F, latent, latent)
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
