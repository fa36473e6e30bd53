"""TEST INFRASTRUCTURE — CPU restatement of the reference's denoising loop (only tests/, smoke() and bench's cpu_baseline leg may
import oracle/).

Follows ``animatediff/pipelines/pipeline.py:1003-1031`` statement by statement with plain torch ops, and restates the
third-party ``diffusers==0.28.0`` ``DDIMScheduler`` (absent from /root/reference and from this image) as ``inference.py:61`` /
``configs/inference/inference.yaml:36-42`` configure it.  PARITY UNPINNED for the scheduler part: no diffusers install or
golden vectors exist offline; the only known answers are structural (leading spacing with steps_offset 1 ends at t = 1, the
last step uses final_alpha_cumprod = 1 and therefore returns the predicted x0, alphas_cumprod[0] = 1 - beta_start).

FreeInit (``pipeline.py:987-999``) is diffusers' ``FreeInitMixin`` (``pipelines/free_init_utils.py`` v0.28.0), also third-party and
absent: restated below method by method (``_get_free_init_freq_filter`` with its Python triple loop, ``_apply_freq_filter``,
``_apply_free_init``), PARITY UNPINNED likewise; known answers are structural (mask = 1 at the shifted DC bin, all-pass mask returns
the re-noised latents, zero mask returns the fresh noise).  The reference-OWNED part of the loop (CFG order and combine, camera
doubling, first-frame re-pin, which frames FreeInit touches) is pinned separately: tests/golden/pipeline_loop.npz holds outputs of the
reference's own loop statement run with these restated third-party pieces (tests/golden/make_pipeline_loop_goldens.py)."""
import math

import torch


class DDIMRef:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.T = num_train_timesteps
        self.steps_offset = steps_offset
        self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)                      # set_alpha_to_one=True (default)

    def set_timesteps(self, n):
        self.n = n
        ratio = self.T // n
        self.timesteps = (torch.arange(0, n) * ratio).round().flip(0).long() + self.steps_offset       # "leading"

    def add_noise(self, original_samples, noise, timesteps):              # DDIMScheduler.add_noise
        acp = self.alphas_cumprod.to(dtype=original_samples.dtype)
        sa = (acp[timesteps] ** 0.5).flatten()
        so = ((1 - acp[timesteps]) ** 0.5).flatten()
        while sa.dim() < original_samples.dim():
            sa, so = sa.unsqueeze(-1), so.unsqueeze(-1)
        return sa * original_samples + so * noise

    def step(self, eps, t, sample):                                       # eta = 0, epsilon prediction, no clipping
        prev = t - self.T // self.n
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = (sample - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


@torch.no_grad()
def denoise_loop_ref(unet, latents, first_frame_latents, prompt_embeds, image_embeds, camera, num_inference_steps=25,
                     guidance_scale=7.5, i2v_cond_time_zero=False):
    sched = DDIMRef()
    sched.set_timesteps(num_inference_steps)
    for t in sched.timesteps:
        latent_model_input = torch.cat([latents] * 2)                                              # pipeline.py:1006
        noise_pred = unet(latent_model_input, t, encoder_hidden_states=prompt_embeds,              # :1010-1020
                          camera=torch.cat([camera] * 2), added_cond_kwargs={"image_embeds": image_embeds},
                          i2v_cond_time_zero=i2v_cond_time_zero).sample
        noise_pred_uncond, noise_pred_text = noise_pred.chunk(2)                                   # :1023-1025
        noise_pred = noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)
        latents = sched.step(noise_pred, int(t), latents)                                          # :1028
        latents = torch.cat([first_frame_latents, latents[:, :, 1:]], dim=2)                       # :1031
    return latents


class FreeInitRef:
    """diffusers FreeInitMixin as ``inference.py:245`` enables it (method, num_iters; order 4, both stop frequencies 0.25,
    use_fast_sampling False)."""

    def __init__(self, scheduler, method="butterworth", order=4, spatial_stop_frequency=0.25, temporal_stop_frequency=0.25):
        self.scheduler, self.method, self.order = scheduler, method, order
        self.ss, self.ts = spatial_stop_frequency, temporal_stop_frequency
        self.initial_noise = None

    def freq_filter(self, shape):                                         # _get_free_init_freq_filter
        time, height, width = shape[-3], shape[-2], shape[-1]
        mask = torch.zeros(shape)
        if self.ss == 0 or self.ts == 0:
            return mask
        if self.method == "butterworth":
            retrieve = lambda x: 1 / (1 + (x / self.ss ** 2) ** self.order)
        elif self.method == "gaussian":
            retrieve = lambda x: math.exp(-1 / (2 * self.ss ** 2) * x)
        elif self.method == "ideal":
            retrieve = lambda x: 1 if x <= self.ss * 2 else 0
        else:
            raise NotImplementedError(self.method)
        for t in range(time):
            for h in range(height):
                for w in range(width):
                    d_square = (((self.ss / self.ts) * (2 * t / time - 1)) ** 2 + (2 * h / height - 1) ** 2 + (2 * w / width - 1) ** 2)
                    mask[..., t, h, w] = retrieve(d_square)
        return mask

    @staticmethod
    def apply_freq_filter(x, noise, low_pass_filter):                     # _apply_freq_filter
        import torch.fft as fft
        x_freq = fft.fftshift(fft.fftn(x, dim=(-3, -2, -1)), dim=(-3, -2, -1))
        noise_freq = fft.fftshift(fft.fftn(noise, dim=(-3, -2, -1)), dim=(-3, -2, -1))
        high_pass_filter = 1 - low_pass_filter
        x_freq_mixed = x_freq * low_pass_filter + noise_freq * high_pass_filter
        x_freq_mixed = fft.ifftshift(x_freq_mixed, dim=(-3, -2, -1))
        return fft.ifftn(x_freq_mixed, dim=(-3, -2, -1)).real

    def apply(self, latents, it, generator):                              # _apply_free_init (use_fast_sampling False)
        if it == 0:
            self.initial_noise = latents.detach().clone()
            return latents
        dtype = latents.dtype
        filt = self.freq_filter((1, *latents.shape[1:]))
        t = torch.full((latents.shape[0],), self.scheduler.T - 1).long()
        z_t = self.scheduler.add_noise(latents, self.initial_noise, t).to(torch.float32)
        z_rand = torch.randn(latents.shape, generator=generator, dtype=torch.float32)
        return self.apply_freq_filter(z_t, z_rand, filt).to(dtype)


@torch.no_grad()
def denoise_free_init_ref(unet, latents, first_frame_latents, prompt_embeds, image_embeds, camera, num_iters=3, generator=None,
                          **loop_kwargs):
    fi = FreeInitRef(DDIMRef())
    for it in range(num_iters):                                                                   # pipeline.py:988
        rest = fi.apply(latents[:, :, 1:], it, generator)                                         # :990-992
        latents = torch.cat([first_frame_latents, rest], dim=2)                                   # :999
        latents = denoise_loop_ref(unet, latents, first_frame_latents, prompt_embeds, image_embeds, camera, **loop_kwargs)
    return latents
