"""TEST INFRASTRUCTURE — CPU restatement of the reference's denoising loop (only tests/, smoke() and bench's cpu_baseline leg may
import oracle/).

Follows ``animatediff/pipelines/pipeline.py:1003-1031`` statement by statement with plain torch ops, and restates the
third-party ``diffusers==0.28.0`` ``DDIMScheduler`` (absent from /root/reference and from this image) as ``inference.py:61`` /
``configs/inference/inference.yaml:36-42`` configure it.  PARITY UNPINNED for the scheduler part: no diffusers install or
golden vectors exist offline; the only known answers are structural (leading spacing with steps_offset 1 ends at t = 1, the
last step uses final_alpha_cumprod = 1 and therefore returns the predicted x0, alphas_cumprod[0] = 1 - beta_start)."""
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
