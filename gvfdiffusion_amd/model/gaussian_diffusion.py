"""The part of the reference's guided-diffusion heritage the inference path needs: the beta schedule
that feeds NoiseScheduleVP (inference_dpm_latent.py:75,156 -> utils/script_util.py:7-61 ->
model/gaussian_diffusion.py:35-89).  Training losses / respacing are out of scope."""
import math

import numpy as np


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """beta_i = min(1 - abar((i+1)/T) / abar(i/T), max_beta), float64."""
    T = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), max_beta) for i in range(T)])


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, beta_start=0.0001, beta_end=0.02):
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * beta_start, scale * beta_end, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


class GaussianDiffusion:
    """Carrier of `.betas` (the only attribute inference reads, inference_dpm_latent.py:156)."""

    def __init__(self, betas, predict_type="eps", rescale_timesteps=False):
        self.betas = np.asarray(betas, dtype=np.float64)
        self.num_timesteps = int(self.betas.shape[0])
        self.predict_type = predict_type
        self.rescale_timesteps = rescale_timesteps


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear", use_kl=False,
                              predict_type="eps", predict_xstart=False, rescale_timesteps=False,
                              rescale_learned_sigmas=False, timestep_respacing="", beta_start=0.0001, beta_end=0.02,
                              min_snr=False):
    """Same keyword surface as utils/script_util.py:7-23 (configs/diffusion.yml `diffusion:` splats into it)."""
    if predict_type not in ("eps", "xstart", "v"):
        raise ValueError(f"Unknown predict_type for diffusion model: {predict_type}")
    betas = get_named_beta_schedule(noise_schedule, steps, beta_start, beta_end)
    # The reference wraps the schedule in SpacedDiffusion (model/respace.py:120-134), which re-derives
    # the betas of the retained timesteps from the cumulative products: beta_i = 1 - abar_i / abar_{i-1}.
    # With every step retained that is the same schedule up to float64 rounding; reproduce it bit for bit.
    if timestep_respacing not in ("", None) and list(timestep_respacing) != [steps]:
        raise NotImplementedError("timestep respacing is a training/ancestral-sampling feature (out of scope)")
    alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
    new_betas, last = [], 1.0
    for acp in alphas_cumprod:
        new_betas.append(1 - acp / last)
        last = acp
    return GaussianDiffusion(np.array(new_betas), predict_type=predict_type, rescale_timesteps=rescale_timesteps)
