"""DDIM / inverse-DDIM schedulers with the interface the reference drives (diffusers 0.25.1 ``DDIMScheduler`` /
``DDIMInverseScheduler`` as constructed at GeoDiffuser/utils/diffusion.py:110 and inversion.py:143):
``set_timesteps``, ``timesteps``, ``alphas_cumprod``, ``config.num_train_timesteps``, ``final_alpha_cumprod``,
``step(model_output, t, sample, eta=0.0)``.  The step arithmetic is the reference's own closed form
(inversion.py:47-65) executed by the ``gd_ddim_step`` HIP kernel.  ``prediction_type="v_prediction"`` (not in the reference: its
README.md:61 lists v-prediction models as to do) runs the same step on x0 / eps recovered from v (``gd_ddim_step_v``), as needed by
SD2.1-768 (BASELINE configs[3]).

Parity note: diffusers is absent, so the timestep table ('leading' spacing, steps_offset 0) is unpinned.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from . import ops


def make_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class _Out(dict):
    @property
    def prev_sample(self):
        return self["prev_sample"]

    def __getitem__(self, k):
        if isinstance(k, int):
            return list(self.values())[k]
        return super().__getitem__(k)


class DDIMScheduler:
    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                 set_alpha_to_one=False, num_train_timesteps=1000, prediction_type="epsilon"):
        if beta_schedule != "scaled_linear" or clip_sample or prediction_type not in ("epsilon", "v_prediction"):
            raise NotImplementedError("only the configuration the reference builds (diffusion.py:110) is supported, plus "
                                      "prediction_type='v_prediction' (SD2.1-768, BASELINE configs[3])")
        self._v = prediction_type == "v_prediction"
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      prediction_type=prediction_type)
        self.alphas_cumprod = make_alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        self._ac = self.alphas_cumprod.tolist()                      # host copy: the step takes its alphas by value
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.initial_alpha_cumprod = self.final_alpha_cumprod
        self._edge = 1.0 if set_alpha_to_one else self._ac[0]
        self.num_inference_steps = None
        self.timesteps = None
        self.init_noise_sigma = 1.0
        self.order = 1

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _alpha(self, t: int) -> float:
        return self._ac[t] if t >= 0 else self._edge

    def step(self, model_output, timestep, sample, eta: float = 0.0, eps_cond=None, guidance_scale: float = 1.0, **kw):
        """x_{t-d} from x_t (inversion.py:47-55).  ``eps_cond``/``guidance_scale``: optional fused CFG combine."""
        if eta != 0.0:
            raise NotImplementedError("eta must be 0 (diffusion.py:52)")
        t = int(timestep)
        tp = t - self.config.num_train_timesteps // self.num_inference_steps
        dt = sample.dtype
        eps = model_output.to(dt).contiguous()
        epc = None if eps_cond is None else eps_cond.to(dt).contiguous()
        out = ops.ddim_step(sample.contiguous(), eps, epc, float(guidance_scale), self._alpha(t), self._alpha(tp), v_prediction=self._v)
        return _Out(prev_sample=out)


class DDIMInverseScheduler(DDIMScheduler):
    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round().copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, eps_cond=None, guidance_scale: float = 1.0, return_dict=True, **kw):
        """x_t from x_{t-d} (inversion.py:57-65)."""
        t = int(timestep)
        tc = min(t - self.config.num_train_timesteps // self.num_inference_steps, self.config.num_train_timesteps - 1)
        dt = sample.dtype
        eps = model_output.to(dt).contiguous()
        epc = None if eps_cond is None else eps_cond.to(dt).contiguous()
        out = ops.ddim_step(sample.contiguous(), eps, epc, float(guidance_scale), self._alpha(tc), self._alpha(t), v_prediction=self._v)
        return _Out(prev_sample=out) if return_dict else (out,)
