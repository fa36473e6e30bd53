"""The denoising loop around the UNet — SURVEY.md §8(f) item 1, the first "next" row after the UNet forward.

Host-side mirror of ``animatediff/pipelines/pipeline.py:1003-1031`` (step 8 of ``AnimationPipeline.__call__``) for the
released configuration (``configs/inference/inference.yaml:36-47``: DDIM, 25 steps, guidance 7.5): per step

    latent_model_input = cat([latents] * 2)                      pipeline.py:1006-1007 (scale_model_input is the identity for DDIM)
    noise_pred = unet(latent_model_input, t, ...).sample         :1010-1020
    CFG combine, scheduler.step, first-frame re-pin              :1023-1031  -> ONE kernel, a3d_cfg_ddim_step_f32

The scheduler is diffusers' ``DDIMScheduler`` as ``inference.py:61`` builds it (third-party, not in /root/reference; restated
from its published algorithm): linear betas, ``timestep_spacing="leading"`` (default), ``steps_offset=1``,
``set_alpha_to_one=True``, epsilon prediction, eta = 0, no clipping.  FreeInit (pipeline.py:989-999) and the step
callbacks are not covered.  The latents stay fp32 on the device for the whole loop; nothing synchronises with the host.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch


def ddim_schedule(num_inference_steps: int, num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                  beta_end: float = 0.012, steps_offset: int = 1) -> Tuple[List[int], torch.Tensor]:
    """(timesteps in loop order, alphas_cumprod[num_train_timesteps] float64).
    DDIMScheduler.__init__ (beta_schedule="linear") and set_timesteps (timestep_spacing="leading")."""
    if not 0 < num_inference_steps <= num_train_timesteps:
        raise ValueError(f"num_inference_steps={num_inference_steps} must be in 1..{num_train_timesteps}")
    betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)      # diffusers builds them in fp32
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).double()
    ratio = num_train_timesteps // num_inference_steps
    timesteps = [i * ratio + steps_offset for i in range(num_inference_steps)][::-1]
    return timesteps, alphas_cumprod


def ddim_alphas(t: int, alphas_cumprod: torch.Tensor, num_inference_steps: int, num_train_timesteps: int = 1000) -> Tuple[float, float]:
    """(alpha_prod_t, alpha_prod_t_prev) of DDIMScheduler.step: prev = t - T // steps; final_alpha_cumprod = 1 below 0."""
    prev = t - num_train_timesteps // num_inference_steps
    a_t = float(alphas_cumprod[t])
    a_prev = float(alphas_cumprod[prev]) if prev >= 0 else 1.0
    return a_t, a_prev


@torch.no_grad()
def denoise_loop(unet, latents: torch.Tensor, first_frame_latents: torch.Tensor, prompt_embeds: torch.Tensor,
                 image_embeds: torch.Tensor, camera: torch.Tensor, num_inference_steps: int = 25,
                 guidance_scale: float = 7.5, i2v_cond_time_zero: bool = False,
                 scheduler_kwargs: Optional[Dict] = None) -> torch.Tensor:
    """latents [n, 4, F, h, w] fp32 with frame 0 = first_frame_latents [n, 4, 1, h, w]; prompt_embeds [2n, 77, 768] and
    image_embeds [2n, 1024] in (uncond, text) order (pipeline.py:931-937); camera [n, 16] (pipeline.py:984).  Returns the
    final latents.  ``unet`` is an ``animate3d_amd.unet.MVUNetMotionModel`` (its ``ops`` supplies the fused step kernel)."""
    if latents.dtype != torch.float32 or not latents.is_cuda:
        raise ValueError("latents must be a float32 CUDA tensor (the reference keeps fp32 latents through scheduler.step)")
    n = latents.shape[0]
    if prompt_embeds.shape[0] != 2 * n or image_embeds.shape[0] != 2 * n or camera.shape[0] != n:
        raise ValueError("classifier-free guidance batch: prompt_embeds / image_embeds need 2n rows, camera n rows")
    timesteps, acp = ddim_schedule(num_inference_steps, **(scheduler_kwargs or {}))
    T = (scheduler_kwargs or {}).get("num_train_timesteps", 1000)
    cam2 = torch.cat([camera, camera])
    first = first_frame_latents.to(torch.float32).contiguous()
    latents = latents.contiguous()
    added = {"image_embeds": image_embeds}
    for t in timesteps:
        model_in = torch.cat([latents, latents])
        eps = unet(model_in, t, encoder_hidden_states=prompt_embeds, camera=cam2, added_cond_kwargs=added,
                   i2v_cond_time_zero=i2v_cond_time_zero).sample
        a_t, a_prev = ddim_alphas(t, acp, num_inference_steps, T)
        latents = unet.ops.cfg_ddim_step(eps.float().contiguous(), latents, first, guidance_scale, a_t, a_prev)
    return latents
