"""Deterministic stand-ins shared by tests/golden/make_sds_goldens.py (which runs the REFERENCE's SDS step with them) and
tests/test_sds.py (which runs the product's with them): a cheap "UNet" whose output depends on every input the real one gets,
and the two diffusers ``DDIMScheduler`` methods the reference's guidance calls (``add_noise``, ``step(...).pred_original_sample``;
third-party, restated from diffusers 0.28.0 as in oracle/denoise_ref.py)."""
from types import SimpleNamespace

import torch


def stub_unet(sample, timestep, encoder_hidden_states=None, camera=None, added_cond_kwargs=None, i2v_cond_time_zero=False, **_):
    """[V, 4, F, h, w] -> object with .sample of the same shape; every conditioning input and the frame index leave a trace."""
    V, C, Fr, h, w = sample.shape
    x = sample.float()
    t = torch.as_tensor(timestep).float().reshape(-1)
    t = t.expand(V) if t.numel() == 1 else t
    txt = encoder_hidden_states.float().mean(dim=(1, 2))                      # [V]
    cam = camera.float() @ torch.linspace(-1.0, 1.0, camera.shape[1])         # [V]
    img = added_cond_kwargs["image_embeds"].float().mean(dim=1)               # [V]
    per_video = (0.001 * t + 0.5 * txt + 0.25 * cam + 2.0 * img).reshape(V, 1, 1, 1, 1)
    frame = torch.arange(Fr, dtype=torch.float32).reshape(1, 1, Fr, 1, 1) * (0.03 if i2v_cond_time_zero else 0.02)
    y = torch.tanh(0.7 * x) + 0.1 * x.roll(1, dims=1) + per_video + frame + 0.05 * x.mean(dim=2, keepdim=True)
    return SimpleNamespace(sample=y.to(sample.dtype))


class StubDDIM:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps)

    def add_noise(self, original_samples, noise, timesteps):
        acp = self.alphas_cumprod.to(dtype=original_samples.dtype)
        sa, so = (acp[timesteps] ** 0.5).flatten(), ((1 - acp[timesteps]) ** 0.5).flatten()
        while sa.dim() < original_samples.dim():
            sa, so = sa.unsqueeze(-1), so.unsqueeze(-1)
        return sa * original_samples + so * noise

    def step(self, model_output, timestep, sample):
        a = self.alphas_cumprod[timestep]
        return SimpleNamespace(pred_original_sample=(sample - (1 - a) ** 0.5 * model_output) / a ** 0.5)
