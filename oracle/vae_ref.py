"""TEST INFRASTRUCTURE — CPU restatement of the VAE decode the reference calls in `decode_latents`
(animatediff/pipelines/pipeline.py:566-579: latents / scaling_factor -> (b f) c h w -> vae.decode -> b c f h w float32)
and of the VAE encode of `encode_latents` (pipeline.py:540-562: vae.encode(x).latent_dist.sample() * scaling_factor).

The VAE itself is diffusers' `AutoencoderKL` (third-party `diffusers==0.28.0`, requirements.txt:2; loaded from the SD1.5
checkpoint at inference.py:62, absent from /root/reference and from this image), restated here from its published structure
with the SD1.5 VAE configuration: block_out_channels (128, 256, 512, 512), layers_per_block 2 (the decoder uses 3 resnets per
up block), norm_num_groups 32, latent_channels 4, scaling_factor 0.18215, one single-head self-attention (head dim 512) in the
mid block, GroupNorm eps 1e-6, SiLU.  Parameter names are diffusers' (`post_quant_conv`, `decoder.*`), so a real VAE
state dict loads key for key.  The encoder half (`encoder.*`, `quant_conv.*`): conv_in 3 -> 128, four DownEncoderBlock2D of two
resnets each, `Downsample2D(padding=0)` = F.pad(x, (0, 1, 0, 1)) + 3x3 stride-2 conv after the first three, the same mid block,
GroupNorm/SiLU/conv_out to 2 x latent_channels, 1x1 `quant_conv`, `DiagonalGaussianDistribution` (logvar clamped to [-30, 20],
sample = mean + exp(0.5 logvar) * randn).  PARITY UNPINNED: no diffusers install, weights or golden vectors exist offline."""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VAEConfig:
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    latent_channels: int = 4
    out_channels: int = 3
    scaling_factor: float = 0.18215
    attention_head_dim: int = 512          # diffusers: mid-block attention heads = channels // attention_head_dim


class Resnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class MidAttention(nn.Module):
    def __init__(self, c, groups, head_dim):
        super().__init__()
        self.heads = max(1, c // head_dim)
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])

    def forward(self, x):
        B, C, H, W = x.shape
        t = self.group_norm(x).reshape(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        split = lambda z: z.reshape(B, H * W, self.heads, C // self.heads).transpose(1, 2)
        a = F.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(B, H * W, C)
        return x + self.to_out[0](a).transpose(1, 2).reshape(B, C, H, W)


class Upsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Mid(nn.Module):
    def __init__(self, c, groups, head_dim):
        super().__init__()
        self.attentions = nn.ModuleList([MidAttention(c, groups, head_dim)])
        self.resnets = nn.ModuleList([Resnet(c, c, groups), Resnet(c, c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class UpBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, upsample):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.upsamplers = nn.ModuleList([Upsample(cout)]) if upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, boc[-1], 3, padding=1)
        self.mid_block = Mid(boc[-1], g, cfg.attention_head_dim)
        rev = list(reversed(boc))
        self.up_blocks = nn.ModuleList()
        cout = rev[0]
        for i, c in enumerate(rev):
            cin, cout = cout, c
            self.up_blocks.append(UpBlock(cin, cout, cfg.layers_per_block + 1, g, upsample=i != len(rev) - 1))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class VAEDecoderRef(nn.Module):
    def __init__(self, cfg: VAEConfig = VAEConfig()):
        super().__init__()
        self.cfg = cfg
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)
        self.decoder = Decoder(cfg)

    @torch.no_grad()
    def decode(self, z):                      # AutoencoderKL.decode(z).sample
        return self.decoder(self.post_quant_conv(z))

    @torch.no_grad()
    def decode_latents(self, latents):        # pipeline.py:566-579
        latents = 1 / self.cfg.scaling_factor * latents
        b, c, f, h, w = latents.shape
        latents = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        image = self.decode(latents)
        video = image[None, :].reshape((b, f, -1) + image.shape[2:]).permute(0, 2, 1, 3, 4)
        return video.float()


class Downsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, downsample):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.downsamplers = nn.ModuleList([Downsample(cout)]) if downsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.downsamplers is None else self.downsamplers[0](x)


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.out_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cout = boc[0]
        for i, c in enumerate(boc):
            cin, cout = cout, c
            self.down_blocks.append(DownBlock(cin, cout, cfg.layers_per_block, g, downsample=i != len(boc) - 1))
        self.mid_block = Mid(boc[-1], g, cfg.attention_head_dim)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class VAEEncoderRef(nn.Module):
    def __init__(self, cfg: VAEConfig = VAEConfig()):
        super().__init__()
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)

    @torch.no_grad()
    def encode(self, x):                      # AutoencoderKL.encode(x).latent_dist -> (mean, logvar)
        mean, logvar = torch.chunk(self.quant_conv(self.encoder(x)), 2, dim=1)
        return mean, torch.clamp(logvar, -30.0, 20.0)

    @torch.no_grad()
    def encode_latents(self, x, generator=None):      # pipeline.py:556-560
        mean, logvar = self.encode(x)
        sample = mean + torch.exp(0.5 * logvar) * torch.randn(mean.shape, generator=generator, dtype=mean.dtype)
        return sample * self.cfg.scaling_factor


def init_synthetic_weights(m: nn.Module, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * fan_in ** -0.5)
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return m
