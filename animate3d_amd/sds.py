"""The 4D-SDS step around the UNet — SURVEY.md §8(c) "Python caller rows" / §8(f) item 3 (SDS-side glue).

Host-side mirror of ``AnimateMVDiffusionGuidance.compute_mvdream_recon_loss``
(custom/threestudio-animate3d/guidance/animatemv_guidance.py:391-507) for one optimisation step of the 4-D representation:

    latents [(b n f), 4, h, w]  (rendered + VAE-encoded, requires grad)
      -> keep frame 0 clean, diffuse frames 1.. to t                                   :415-430
      -> ONE UNet forward on the CFG-doubled batch in (text, uncond) order              :433-448   (pipeline uses (uncond, text))
      -> eps = eps_text + s * (eps_text - eps_uncond)                                   :451-459   (NOT uncond + s * (...))
      -> x0 reconstruction, optional std rescale against the un-guided reconstruction   :465-488
      -> frame 0 of the target := the input's frame 0; loss = 0.5 * SSE / N * f/(f-1)    :491-503

Everything outside the UNet call is element-wise work on a [(b n f), 4, 32, 32] tensor: torch ops, differentiable w.r.t.
``latents`` exactly where the reference's are (the target is detached).  The scheduler methods the reference calls
(diffusers ``DDIMScheduler.add_noise`` / ``.step(...).pred_original_sample``, third-party) reduce to the two closed forms
below with ``alphas_cumprod`` from ``denoise.ddim_schedule``.  For b > 1 the reference indexes ``alphas_cumprod[t]`` with t of
shape [b] and broadcasts it against ``(b n f) c h w`` (only well-formed for b = 1, its default); here alpha is applied per b.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from .denoise import ddim_schedule, randn_like_reference


def normalize_camera(c2w: torch.Tensor) -> torch.Tensor:
    """[B, 4, 4] camera-to-world -> [B, 16] with the translation on the unit sphere (pipeline.py:176-190, called by
    ``get_camera_cond``, animatemv_guidance.py:345-358, ``camera_condition_type == "rotation"``)."""
    m = c2w.reshape(-1, 4, 4).clone()
    tr = m[:, :3, 3]
    m[:, :3, 3] = tr / (torch.norm(tr, dim=1, keepdim=True) + 1e-8)
    return m.reshape(-1, 16)


def sds_recon_loss(unet, latents: torch.Tensor, t: torch.Tensor, text_embeddings: torch.Tensor, image_embeds: torch.Tensor,
                   c2w: Optional[torch.Tensor] = None, *, n_view: int = 4, n_frame: int = 8, guidance_scale: float = 100.0,
                   recon_std_rescale: float = 0.5, i2v_cond_time_zero: bool = False, alphas_cumprod: Optional[torch.Tensor] = None,
                   noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                   weights_dtype: Optional[torch.dtype] = None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """One SDS reconstruction step.  ``latents`` [(b n f), 4, h, w]; ``t`` [b] long; ``text_embeddings`` [2 b n, L, D] in
    (text, uncond) order; ``image_embeds`` [b n, E] (the unconditional half is zeros, :439); ``c2w`` [(b n f), 4, 4] or None.
    Returns (loss, {"latents_noisy", "noise_pred", "latents_recon"}) like the reference's ``guidance_eval_utils``."""
    n, f = n_view, n_frame
    if latents.shape[0] % (n * f) != 0:
        raise ValueError(f"latents batch {latents.shape[0]} is not a multiple of n_view * n_frame = {n * f}")
    if f < 2:
        raise ValueError("n_frame must be >= 2 (frame 0 is the clean conditioning frame)")
    b = latents.shape[0] // (n * f)
    c, h, w = latents.shape[1:]
    if t.shape != (b,):
        raise ValueError(f"t must have shape [{b}]")
    if text_embeddings.shape[0] != 2 * b * n or image_embeds.shape[0] != b * n:
        raise ValueError("text_embeddings needs 2*b*n rows ((text, uncond) order), image_embeds b*n rows")
    if alphas_cumprod is None:
        alphas_cumprod = ddim_schedule(1)[1].float()
    acp = alphas_cumprod.to(device=latents.device, dtype=latents.dtype)[t.to(latents.device)]        # [b]
    wd = weights_dtype or latents.dtype

    videos = latents.reshape(b, n, f, c, h, w)
    with torch.no_grad():
        rest = videos[:, :, 1:]
        if noise is None:
            noise = randn_like_reference(rest.shape, generator, latents.device, latents.dtype)
        else:                                               # the reference draws it in "b n c f h w" order; accept that layout too
            noise = noise.to(latents)
            if noise.shape == (b, n, c, f - 1, h, w):
                noise = noise.permute(0, 1, 3, 2, 4, 5)
        a6 = acp.reshape(b, 1, 1, 1, 1, 1)
        noisy = torch.cat([videos[:, :, :1], a6.sqrt() * rest + (1 - a6).sqrt() * noise], dim=2)        # b n f c h w
        sample = noisy.permute(0, 1, 3, 2, 4, 5).reshape(b * n, c, f, h, w)                              # (b n) c f h w
        ts = t.reshape(b, 1).expand(b, n).reshape(-1)
        camera = None
        if c2w is not None:
            cam = normalize_camera(c2w.reshape(b, n, f, 4, 4)[:, :, 0].reshape(b * n, 4, 4)).to(latents.dtype)
            camera = torch.cat([cam, cam]).to(wd)
        embeds = torch.cat([image_embeds, torch.zeros_like(image_embeds)]).to(wd)
        eps = unet(torch.cat([sample, sample]).to(wd), torch.cat([ts, ts]).to(wd), encoder_hidden_states=text_embeddings.to(wd),
                   camera=camera, added_cond_kwargs={"image_embeds": embeds}, i2v_cond_time_zero=i2v_cond_time_zero).sample.to(latents.dtype)
        frames = lambda e: e.permute(0, 2, 1, 3, 4).reshape(b * n * f, c, h, w)                          # (b n) c f h w -> (b n f) c h w
        eps_text, eps_uncond = eps.chunk(2)
        eps_text, eps_uncond = frames(eps_text), frames(eps_uncond)
        eps_cfg = eps_text + guidance_scale * (eps_text - eps_uncond)
        noisy_flat = noisy.reshape(b * n * f, c, h, w)
        a4 = acp.repeat_interleave(n * f).reshape(-1, 1, 1, 1)
        x0 = lambda e: (noisy_flat - (1 - a4).sqrt() * e) / a4.sqrt()
        recon = x0(eps_cfg)
        if recon_std_rescale > 0:
            rest_std = lambda z: z.reshape(b, n, f, c, h, w)[:, :, 1:].std(dim=[1, 2, 3, 4, 5], keepdim=True)   # frames 1.., per b
            factor = (rest_std(x0(eps_text)) + 1e-8) / (rest_std(recon) + 1e-8)
            adjusted = recon * factor.reshape(b).repeat_interleave(n * f).reshape(-1, 1, 1, 1)
            recon = recon_std_rescale * adjusted + (1 - recon_std_rescale) * recon
        recon = recon.reshape(b * n, f, c, h, w)
        recon = torch.cat([videos.detach().reshape(b * n, f, c, h, w)[:, :1], recon[:, 1:]], dim=1).reshape(b * n * f, c, h, w)
    loss = 0.5 * ((latents - recon) ** 2).sum() / latents.shape[0] * f / (f - 1)
    return loss, {"latents_noisy": noisy_flat, "noise_pred": eps_cfg, "latents_recon": recon}
