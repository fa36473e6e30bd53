"""The denoising loop around the UNet — SURVEY.md §8(f) item 1, the first "next" row after the UNet forward.

Host-side mirror of ``animatediff/pipelines/pipeline.py:1003-1031`` (step 8 of ``AnimationPipeline.__call__``) for the
released configuration (``configs/inference/inference.yaml:36-47``: DDIM, 25 steps, guidance 7.5): per step

    latent_model_input = cat([latents] * 2)                      pipeline.py:1006-1007 (scale_model_input is the identity for DDIM)
    noise_pred = unet(latent_model_input, t, ...).sample         :1010-1020
    CFG combine, scheduler.step, first-frame re-pin              :1023-1031  -> ONE kernel, a3d_cfg_ddim_step_f32

The scheduler is diffusers' ``DDIMScheduler`` as ``inference.py:61`` builds it (third-party, not in /root/reference; restated
from its published algorithm): linear betas, ``timestep_spacing="leading"`` (default), ``steps_offset=1``,
``set_alpha_to_one=True``, epsilon prediction, eta = 0, no clipping.  The latents stay fp32 on the device for the whole
loop; nothing synchronises with the host.  The step callbacks are not covered.

FreeInit (``pipeline.py:987-999``; enabled in the released config, ``configs/inference/inference.yaml:27-28`` and
``inference.py:244-245``: butterworth, 3 iterations, no fast sampling) wraps the loop: ``denoise_free_init`` below.  It is
diffusers' ``FreeInitMixin`` (``diffusers/pipelines/free_init_utils.py`` v0.28.0, third-party, restated): after the first pass
the frames 1.. of the result are diffused back to t = T-1 with the ORIGINAL noise, and their high 3-D frequencies are replaced
by those of fresh noise.  Two FFTs of a [n, 4, F-1, h, w] tensor three times per sample: torch.fft (rocFFT) is plumbing here.
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


def randn_like_reference(shape, generator: Optional[torch.Generator], device, dtype: torch.dtype) -> torch.Tensor:
    """diffusers' ``randn_tensor``: draw on the generator's device (a CPU generator is the usual idiom, ``inference.py``
    seeds one) and move; without a generator draw on ``device``."""
    device = torch.device(device)
    gdev = generator.device if generator is not None else device
    return torch.randn(tuple(shape), generator=generator, device=gdev, dtype=dtype).to(device)


def prepare_latents(first_frame_latents: torch.Tensor, num_frames: int, generator: Optional[torch.Generator] = None,
                    device=None, dtype: torch.dtype = torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """Step 5 of ``__call__`` (pipeline.py:950-973, ``i2v_similarity_init`` = None as released): the encoded conditioning images
    ``[n, 4, h, w]`` become frame 0, frames 1.. are N(0, 1) noise times the scheduler's ``init_noise_sigma`` (1 for DDIM), drawn
    as diffusers' ``randn_tensor`` draws them (on the generator's device, then moved).  Returns (latents [n, 4, F, h, w],
    first_frame_latents [n, 4, 1, h, w])."""
    if num_frames < 2:
        raise ValueError("num_frames must be >= 2 (frame 0 is the conditioning frame)")
    device = torch.device(device) if device is not None else first_frame_latents.device
    first = first_frame_latents if first_frame_latents.dim() == 5 else first_frame_latents.unsqueeze(2)
    first = first.to(device=device, dtype=dtype)
    n, c, _, h, w = first.shape
    rest = randn_like_reference((n, c, num_frames - 1, h, w), generator, device, dtype)
    return torch.cat([first, rest], dim=2), first


@torch.no_grad()
def denoise_loop(unet, latents: torch.Tensor, first_frame_latents: torch.Tensor, prompt_embeds: torch.Tensor,
                 image_embeds: torch.Tensor, camera: torch.Tensor, num_inference_steps: int = 25,
                 guidance_scale: float = 7.5, i2v_cond_time_zero: bool = False,
                 scheduler_kwargs: Optional[Dict] = None) -> torch.Tensor:
    """latents [n, 4, F, h, w] fp32 with frame 0 = first_frame_latents [n, 4, 1, h, w]; prompt_embeds [2n, 77, 768] and
    image_embeds [2n, 1024] in (uncond, text) order (pipeline.py:931-937); camera [n, 16] (pipeline.py:984).  Returns the
    final latents.  ``unet`` is an ``animate3d_amd.unet.MVUNetMotionModel`` (its ``ops`` supplies the fused step kernel)."""
    if latents.dtype != torch.float32:
        raise ValueError("latents must be float32 (the reference keeps fp32 latents through scheduler.step)")
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


def free_init_filter(shape, method: str = "butterworth", order: int = 4, spatial_stop_frequency: float = 0.25,
                     temporal_stop_frequency: float = 0.25, device=None) -> torch.Tensor:
    """Low-pass mask over the (fft-shifted) last three dims ``(time, height, width)`` of ``shape``; float32, broadcastable
    ``[1, ..., time, height, width]``.  FreeInitMixin._get_free_init_freq_filter, vectorised (the original fills it with a
    Python triple loop): d^2 = (s/t_stop * (2t/T - 1))^2 + (2y/H - 1)^2 + (2x/W - 1)^2 ; butterworth 1 / (1 + (d^2/s^2)^order)."""
    time, height, width = int(shape[-3]), int(shape[-2]), int(shape[-1])
    lead = (1,) * (len(shape) - 3)
    s, ts = float(spatial_stop_frequency), float(temporal_stop_frequency)
    if s == 0 or ts == 0:
        return torch.zeros(lead + (time, height, width), dtype=torch.float32, device=device)
    ax = lambda n: 2.0 * torch.arange(n, dtype=torch.float64) / n - 1.0
    d2 = ((s / ts) * ax(time))[:, None, None] ** 2 + ax(height)[None, :, None] ** 2 + ax(width)[None, None, :] ** 2
    if method == "butterworth":
        mask = 1.0 / (1.0 + (d2 / s ** 2) ** order)
    elif method == "gaussian":
        mask = torch.exp(-1.0 / (2.0 * s ** 2) * d2)
    elif method == "ideal":
        mask = (d2 <= s * 2).double()
    else:
        raise NotImplementedError(f"FreeInit filter method {method!r} (butterworth, gaussian, ideal)")
    return mask.to(torch.float32).reshape(lead + (time, height, width)).to(device)


def free_init_mix(x: torch.Tensor, noise: torch.Tensor, low_pass_filter: torch.Tensor) -> torch.Tensor:
    """Low 3-D frequencies of ``x`` + high frequencies of ``noise`` (FreeInitMixin._apply_freq_filter); float32 in and out."""
    dims = (-3, -2, -1)
    xf = torch.fft.fftshift(torch.fft.fftn(x, dim=dims), dim=dims)
    nf = torch.fft.fftshift(torch.fft.fftn(noise, dim=dims), dim=dims)
    mixed = xf * low_pass_filter + nf * (1 - low_pass_filter)
    return torch.fft.ifftn(torch.fft.ifftshift(mixed, dim=dims), dim=dims).real


def free_init_renoise(rest_latents: torch.Tensor, initial_noise: torch.Tensor, alpha_prod_T: float, low_pass_filter: torch.Tensor,
                      generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """FreeInitMixin._apply_free_init for iteration > 0: ``scheduler.add_noise(latents, initial_noise, T-1)`` in fp32, fresh
    fp32 noise of the same shape from ``generator``, frequency mix."""
    z_t = (alpha_prod_T ** 0.5) * rest_latents.float() + ((1.0 - alpha_prod_T) ** 0.5) * initial_noise.float()
    z_rand = randn_like_reference(rest_latents.shape, generator, rest_latents.device, torch.float32)
    return free_init_mix(z_t, z_rand, low_pass_filter).to(rest_latents.dtype)


@torch.no_grad()
def denoise_free_init(unet, latents: torch.Tensor, first_frame_latents: torch.Tensor, prompt_embeds: torch.Tensor,
                      image_embeds: torch.Tensor, camera: torch.Tensor, num_iters: int = 3, generator: Optional[torch.Generator] = None,
                      method: str = "butterworth", order: int = 4, spatial_stop_frequency: float = 0.25,
                      temporal_stop_frequency: float = 0.25, loop=None, **loop_kwargs) -> torch.Tensor:
    """``pipeline.py:987-1031`` with ``free_init_enabled``: ``num_iters`` full denoising passes; before pass k > 0 the frames
    1.. are re-initialised from the previous result (only they: ``pipeline.py:990-992``) and the conditioning frame is put back
    (``:999``).  ``loop`` defaults to ``denoise_loop`` (same keyword arguments, e.g. ``num_inference_steps``, ``guidance_scale``)."""
    if num_iters < 1:
        raise ValueError("num_iters must be >= 1")
    loop = loop or denoise_loop
    sk = loop_kwargs.get("scheduler_kwargs") or {}
    T = sk.get("num_train_timesteps", 1000)
    _, acp = ddim_schedule(1, **sk)
    a_T = float(acp[T - 1])
    initial_noise = latents[:, :, 1:].detach().clone()                         # iteration 0 keeps the caller's noise
    filt = None
    for it in range(num_iters):
        if it > 0:
            if filt is None:
                filt = free_init_filter((1,) + tuple(initial_noise.shape[1:]), method, order, spatial_stop_frequency,
                                        temporal_stop_frequency, device=latents.device)
            rest = free_init_renoise(latents[:, :, 1:], initial_noise, a_T, filt, generator)
            latents = torch.cat([first_frame_latents.to(rest.dtype), rest], dim=2)
        latents = loop(unet, latents, first_frame_latents, prompt_embeds, image_embeds, camera, **loop_kwargs)
    return latents
