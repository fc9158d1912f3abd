"""DDIM scheduler with the configuration the reference builds (inference.py:132-136 from
inference_v2.yaml:24-35): scaled-linear betas rescaled to zero terminal SNR, trailing timestep spacing,
v-prediction, eta = 0.  The arithmetic is diffusers==0.29.2 `DDIMScheduler` (absent third-party dependency;
restated from its published behaviour — SURVEY.md Appendix A).

Host side only: the per-step update itself runs in the fused `vx_overlap_ddim_step` kernel, fed by
`step_coefficients(t)`; `step()` is kept as the reference-compatible tensor API.
"""
from types import SimpleNamespace

import numpy as np
import torch


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 timestep_spacing="leading", rescale_betas_zero_snr=False, **unused):
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        if rescale_betas_zero_snr:
            abar_sqrt = torch.cumprod(1.0 - betas, 0).sqrt()
            a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
            abar_sqrt = (abar_sqrt - aT) * (a0 / (a0 - aT))
            abar = abar_sqrt ** 2
            betas = 1 - torch.cat([abar[0:1], abar[1:] / abar[:-1]])
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, 0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, clip_sample=clip_sample,
                                      steps_offset=steps_offset, prediction_type=prediction_type,
                                      timestep_spacing=timestep_spacing)
        if clip_sample:
            raise NotImplementedError("clip_sample=True is not used by V-Express (inference_v2.yaml:28)")
        if prediction_type != "v_prediction":
            raise NotImplementedError("only v_prediction (inference_v2.yaml:31) is built")
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)

    def set_timesteps(self, num_inference_steps, device=None):
        T, sp = self.config.num_train_timesteps, self.config.timestep_spacing
        self.num_inference_steps = num_inference_steps
        if sp == "trailing":
            ts = np.round(np.arange(T, 0, -T / num_inference_steps)).astype(np.int64) - 1
        elif sp == "leading":
            ts = (np.arange(0, num_inference_steps) * (T // num_inference_steps)).round()[::-1].astype(np.int64)
            ts = ts + self.config.steps_offset
        elif sp == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].astype(np.int64)
        else:
            raise ValueError(sp)
        self.timesteps = torch.from_numpy(ts.copy())

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _alphas(self, t):
        t = int(t)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return a, a_prev

    def step_coefficients(self, t):
        """(sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)) as fp32-rounded Python floats: exactly the
        four scalars of DDIMScheduler.step for v-prediction, eta=0."""
        a, a_prev = self._alphas(t)
        return (float(a ** 0.5), float((1 - a) ** 0.5), float(a_prev ** 0.5), float((1 - a_prev) ** 0.5))

    def step(self, model_output, timestep, sample, eta=0.0, **unused):
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is not used by V-Express")
        sa, s1a, sap, s1ap = self.step_coefficients(timestep)
        x0 = sa * sample - s1a * model_output
        eps = sa * model_output + s1a * sample
        return SimpleNamespace(prev_sample=sap * x0 + s1ap * eps, pred_original_sample=x0)
