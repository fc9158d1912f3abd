"""Generate tests/golden/*.pt by running the REFERENCE ITSELF (imported unmodified over the diffusers
stand-in) on the seeded synthetic weights/inputs.  Run in the dev container:  python tests/make_golden.py
The GPU box has no /root/reference; it checks oracle/ and the HIP path against these committed outputs."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import cases  # noqa: E402
import refharness as H  # noqa: E402
from v_express_amd import synth  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count())
    out = os.path.join(HERE, "golden")
    os.makedirs(out, exist_ok=True)
    built = {}
    for name, (kw, F, h, w, t) in cases.FORWARD_CASES.items():
        key = tuple(kw["block_out_channels"])
        if key not in built:
            cfg = cases.unet_cfg(kw)
            built = {key: H.build_reference_unets(cfg)}        # keep only one model pair alive
        unet, refnet = built[key]
        inp = synth.synthetic_inputs(cases.unet_cfg(kw), F, h, w)
        pred, banks = H.reference_unet_forward(unet, refnet, inp, t, cases.W_REF, cases.W_AUD)
        g = dict(pred=pred.clone(),
                 bank_stats={k: torch.stack([v.mean(), v.abs().mean(), v.flatten()[::97].sum()])
                             for k, v in banks.items()})
        torch.save(g, os.path.join(out, f"forward_{name}.pt"))
        print(name, pred.shape, float(pred.std()))
    kw = cases.SMALL
    cfg = cases.unet_cfg(kw)
    unet, refnet = H.build_reference_unets(cfg)
    vcfg = synth.VaeConfig(**cases.SMALL_VAE)
    vae = H.build_reference_vae(vcfg)
    for name, (F, cf, co, steps) in cases.PIPELINE_CASES.items():
        inp = synth.synthetic_inputs(cfg, F, 8, 8)
        video, trace = H.reference_pipeline_run(unet, refnet, vae, inp, F, steps, cases.GUIDANCE, cf, co,
                                                cases.W_REF, cases.W_AUD, 64, 64)
        g = dict(latents=trace[-1].clone(), latents_step0=trace[0].clone(),
                 video_f16=video.to(torch.float16))
        torch.save(g, os.path.join(out, f"pipeline_{name}.pt"))
        print(name, video.shape, float(video.mean()))
    # window lists straight from the reference's pipelines/context.py
    from pipelines.context import uniform
    wins = {}
    for (F, cs, co) in [(16, 16, 4), (64, 16, 4), (124, 16, 4), (128, 16, 4), (100, 24, 4), (924, 24, 4), (11, 4, 2),
                        (7, 8, 2), (40, 24, 4)]:
        wins[(F, cs, co)] = list(uniform(step=0, num_frames=F, context_size=cs, context_stride=1,
                                         context_overlap=co, closed_loop=False))
    torch.save(wins, os.path.join(out, "windows.pt"))
    # DDIM constants straight from the stand-in scheduler the reference pipeline drives
    import diffusers
    import ref_import as R
    s = diffusers.DDIMScheduler(**R.NOISE_SCHEDULER_KWARGS)
    s.set_timesteps(25)
    torch.save(dict(timesteps=s.timesteps.clone(), alphas_cumprod=s.alphas_cumprod.clone()),
               os.path.join(out, "ddim.pt"))


if __name__ == "__main__":
    main()
