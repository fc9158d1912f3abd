"""Oracle restatement of the sliding-window scheduler, the DDIM update and the mean-overlap loop
(TEST INFRASTRUCTURE).

* `uniform_windows`  — pipelines/context.py:30-59 (`uniform`) with `ordered_halving` (:22-27).
* `DDIM`             — diffusers==0.29.2 DDIMScheduler as configured by inference_v2.yaml:24-35
                        (absent third-party dependency, restated).
* `mean_overlap`     — pipelines/v_express_pipeline.py:498-500 (coverage counts) and :526-583 (loop).
"""
import numpy as np
import torch


def ordered_halving(val):
    """pipelines/context.py:22-27."""
    return int(f"{val:064b}"[::-1], 2) / (1 << 64)


def uniform_windows(num_frames, context_size, context_overlap, step=0, context_stride=1, closed_loop=False):
    """pipelines/context.py:30-59 -> list of frame-index lists."""
    if num_frames <= context_size:
        return [list(range(num_frames))]
    out = []
    context_stride = min(context_stride, int(np.ceil(np.log2(num_frames / context_size))) + 1)
    for context_step in 1 << np.arange(context_stride):
        context_step = int(context_step)
        pad = int(round(num_frames * ordered_halving(step)))
        for j in range(int(ordered_halving(step) * context_step) + pad,
                       num_frames + pad + (0 if closed_loop else -context_overlap),
                       context_size * context_step - context_overlap):
            nxt = []
            for e in range(j, j + context_size * context_step, context_step):
                if e >= num_frames:
                    e = num_frames - 2 - e % num_frames
                nxt.append(e)
            out.append(nxt)
    return out


class DDIM:
    """DDIMScheduler(beta_start=0.00085, beta_end=0.012, 'scaled_linear', clip_sample=False, steps_offset=1,
    prediction_type='v_prediction', rescale_betas_zero_snr=True, timestep_spacing='trailing'), eta=0."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        # rescale_zero_terminal_snr
        abar_sqrt = torch.cumprod(1.0 - betas, 0).sqrt()
        a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
        abar_sqrt = (abar_sqrt - aT) * (a0 / (a0 - aT))
        abar = abar_sqrt ** 2
        alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
        self.alphas_cumprod = torch.cumprod(alphas, 0)
        self.T = num_train_timesteps
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.n = n
        self.timesteps = (np.round(np.arange(self.T, 0, -self.T / n)).astype(np.int64) - 1).tolist()
        return self.timesteps

    def step(self, v, t, x):
        prev_t = t - self.T // self.n
        a = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        x0 = a ** 0.5 * x - (1 - a) ** 0.5 * v
        eps = a ** 0.5 * v + (1 - a) ** 0.5 * x
        return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


def mean_overlap(unet_fn, latents, timesteps, ddim, windows, guidance_scale, kps_feature, audio_embeddings,
                 callback=None):
    """pipelines/v_express_pipeline.py:526-583.

    unet_fn(input_latents[b,4,f,h,w], t, ehs[b*f,5,768], kps[b,320,f,h,w]) -> [b,4,f,h,w]
    latents [1,4,F,h,w]; kps_feature [b,320,F,h,w]; audio_embeddings [b,F,5,768] with b = 2 (uncond, cond) under
    classifier-free guidance (guidance_scale > 1, :443) and b = 1 without.  Returns final latents.
    """
    do_cfg = guidance_scale > 1.0
    latents = latents.clone()
    F_ = latents.shape[2]
    num_frame_context = torch.zeros(F_, dtype=torch.long)
    for ctx in windows:
        num_frame_context[ctx] += 1                                       # :498-500 (duplicates count once)
    for i, t in enumerate(timesteps):
        counter = torch.zeros(F_, dtype=torch.long)
        noise_preds = [None] * F_
        for ctx in windows:
            kps = kps_feature[:, :, ctx]
            aud = audio_embeddings[:, ctx]
            aud = aud.reshape(-1, aud.shape[-2], aud.shape[-1])
            inp = latents[:, :, ctx].repeat(2 if do_cfg else 1, 1, 1, 1, 1)    # :539
            pred = unet_fn(inp, t, aud, kps)
            if do_cfg:
                u, c = pred.chunk(2)
                pred = u + guidance_scale * (c - u)                        # :548-550
            counter[ctx] += 1
            pred = pred / num_frame_context[ctx][None, None, :, None, None]   # :553
            ids, preds = [], []
            for li, fi in enumerate(ctx):                                 # :556-564
                if noise_preds[fi] is None:
                    noise_preds[fi] = pred[:, :, li].clone()
                else:
                    noise_preds[fi] = noise_preds[fi] + pred[:, :, li]
                if counter[fi] == num_frame_context[fi]:
                    ids.append(fi)
                    preds.append(noise_preds[fi])
                    noise_preds[fi] = None
            if ids:
                step_pred = torch.stack(preds, dim=2)
                latents[:, :, ids] = ddim.step(step_pred, t, latents[:, :, ids])   # :565-572
        if callback is not None:
            callback(i, t, latents)
    return latents
