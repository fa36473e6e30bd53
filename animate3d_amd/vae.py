"""MI355X-native VAE decode and encode — SURVEY.md §8(f) item 2, the next row after the denoising loop.

Drop-in for what ``AnimationPipeline.decode_latents`` needs (animatediff/pipelines/pipeline.py:566-579):
``latents / scaling_factor -> (b f) c h w -> vae.decode(...).sample -> b c f h w float32``.  The VAE is diffusers'
``AutoencoderKL`` in its SD1.5 configuration (loaded at inference.py:62; third-party, restated from its published structure —
see oracle/vae_ref.py).  Parameter names are diffusers' (``post_quant_conv.*``, ``decoder.*``), so
``vae.state_dict()`` of a real checkpoint loads key for key (``strict=False`` skips the encoder, which the denoise path
never runs per step).

Everything arithmetic goes through the same C-ABI as the UNet (include/animate3d_hip.h) on token-major NHWC rows
``[(b f) h w, C]`` in bf16: 3x3 convs (the persistent LDS-DMA kernel where the shape qualifies, nearest-2x upsample folded into
the gather), GroupNorm+SiLU, 1x1 shortcut GEMMs with the residual add as epilogue; the single-head 512-wide mid-block
attention is four GEMMs + a row softmax on fp32 logits (S is never rounded to bf16; the V bias is added after P V because
softmax rows sum to one).  ``post_quant_conv`` and the ``1 / scaling_factor`` are one 4-channel fp32 kernel.  No fallback.

``AutoencoderKLEncoder`` is the other half (``encoder.*``, ``quant_conv.*``) for ``AnimationPipeline.encode_latents``
(pipeline.py:540-562: ``vae.encode(images).latent_dist.sample() * scaling_factor`` gives the conditioning first-frame latents).
Same kernels; diffusers' ``Downsample2D(padding=0)`` pads right/bottom only before its stride-2 conv — that equals the pad-1
stride-2 conv of the C-ABI on the spatially flipped image with the flipped filter, flipped back (even sizes), so no new kernel.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .denoise import randn_like_reference
from .hip_ops import on_model_device


@dataclass
class VAEConfig:
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    latent_channels: int = 4
    out_channels: int = 3
    scaling_factor: float = 0.18215
    attention_head_dim: int = 512


class _Resnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None


class _Attention(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])


class _Upsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([_Attention(c, groups)])
        self.resnets = nn.ModuleList([_Resnet(c, c, groups), _Resnet(c, c, groups)])


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, upsample):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.upsamplers = nn.ModuleList([_Upsample(cout)]) if upsample else None


class _Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, boc[-1], 3, padding=1)
        self.mid_block = _Mid(boc[-1], g)
        rev = list(reversed(boc))
        self.up_blocks = nn.ModuleList()
        cout = rev[0]
        for i, c in enumerate(rev):
            cin, cout = cout, c
            self.up_blocks.append(_UpBlock(cin, cout, cfg.layers_per_block + 1, g, upsample=i != len(rev) - 1))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)


class _Downsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, downsample):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.downsamplers = nn.ModuleList([_Downsample(cout)]) if downsample else None


class _Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.out_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cout = boc[0]
        for i, c in enumerate(boc):
            cin, cout = cout, c
            self.down_blocks.append(_DownBlock(cin, cout, cfg.layers_per_block, g, downsample=i != len(boc) - 1))
        self.mid_block = _Mid(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)


class _VAEHalf(nn.Module):
    """What the two halves share: op-set plumbing, weight packing and the resnet / mid-attention forward pieces."""

    def __init__(self, config: Optional[VAEConfig], ops):
        super().__init__()
        cfg = config or VAEConfig()
        if cfg.block_out_channels[-1] // cfg.attention_head_dim > 1:
            raise ValueError("only the single-head mid-block attention of the SD VAE is implemented")
        self.config = cfg
        self._ops = ops
        self._ops_auto = False
        self._packed = None

    # ------------------------------------------------------------------ plumbing
    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def ops(self):
        if self._ops is None:
            from .hip_ops import HipOps          # raises without an MI355X or without the built library
            self._ops = HipOps(self.device, torch.float16 if self.dtype == torch.float16 else torch.bfloat16)
            self._ops_auto = True
        return self._ops

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._packed = None
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _apply(self, fn, *a, **kw):
        self._packed = None
        if getattr(self, "_ops_auto", False):
            self._ops, self._ops_auto = None, False
        return super()._apply(fn, *a, **kw)

    def init_synthetic(self, seed: int = 0):
        g = torch.Generator(device=self.device).manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if p.dim() >= 2:
                    p.copy_(((torch.rand(p.shape, generator=g, device=p.device) * 2 - 1) * p[0].numel() ** -0.5).to(p.dtype))
                elif "norm" in name and name.endswith("weight"):
                    p.copy_((1 + 0.1 * torch.randn(p.shape, generator=g, device=p.device)).to(p.dtype))
                else:
                    p.copy_((0.05 * torch.randn(p.shape, generator=g, device=p.device)).to(p.dtype))
        self._packed = None
        return self

    # ------------------------------------------------------------------ weight packing
    def _w(self, t):
        return t.detach().to(self.ops.act_dtype).contiguous()

    def _f(self, t):
        return None if t is None else t.detach().float().contiguous()

    def _conv_w(self, conv, pad_out_to: int = 0):          # [Cout, Cin, 3, 3] -> [Cout, (ky, kx, ci)]
        w = conv.weight.detach().float().permute(0, 2, 3, 1).reshape(conv.weight.shape[0], -1)
        b = conv.bias.detach().float()
        if pad_out_to > w.shape[0]:                        # conv_out: 3 -> 4 output channels (the kernels need N % 4 == 0)
            w = torch.cat([w, w.new_zeros(pad_out_to - w.shape[0], w.shape[1])])
            b = torch.cat([b, b.new_zeros(pad_out_to - b.shape[0])])
        return self._w(w), b.contiguous()

    def _pack_resnet(self, r: _Resnet):
        sc = None
        if r.conv_shortcut is not None:
            sc = (self._w(r.conv_shortcut.weight.detach().reshape(r.conv_shortcut.weight.shape[0], -1)), self._f(r.conv_shortcut.bias))
        return SimpleNamespace(n1=(self._f(r.norm1.weight), self._f(r.norm1.bias)), c1=self._conv_w(r.conv1),
                               n2=(self._f(r.norm2.weight), self._f(r.norm2.bias)), c2=self._conv_w(r.conv2), sc=sc)

    # ------------------------------------------------------------------ forward pieces (rows = [B*H*W, C] bf16)
    def _resnet(self, x, B, H, W, pk):
        ops, g = self.ops, self.config.norm_num_groups
        h = ops.group_norm(x, B, H * W, pk.n1[0], pk.n1[1], g, 1e-6, True)
        h, _, _ = ops.conv3x3(h, B, H, W, pk.c1[0], pk.c1[1])
        h = ops.group_norm(h, B, H * W, pk.n2[0], pk.n2[1], g, 1e-6, True)
        sc = x if pk.sc is None else ops.gemm(x, pk.sc[0], pk.sc[1])
        out, _, _ = ops.conv3x3(h, B, H, W, pk.c2[0], pk.c2[1], residual=sc)
        return out

    def _mid_attention(self, x, B, H, W, pk):
        """diffusers Attention (AttnProcessor, one head of dim C, residual_connection=True) on the GroupNorm'ed tokens."""
        ops, g = self.ops, self.config.norm_num_groups
        L, C = H * W, x.shape[1]
        t = ops.group_norm(x, B, L, pk.gn[0], pk.gn[1], g, 1e-6, False)
        q = ops.gemm(t, pk.q[0], pk.q[1])
        k = ops.gemm(t, pk.k[0], pk.k[1])
        a = ops.empty(B * L, C)
        Lp = -(-L // 64) * 64                                # the P V contraction runs over L: the GEMM needs a multiple of 64
        if Lp != L:                                          # other latent sizes: zero-padded P columns / V^T columns
            p_pad, vt_pad = ops.empty(L, Lp).zero_(), ops.empty(C, Lp).zero_()
        for b in range(B):                                   # per image: L x L logits in fp32, never rounded to bf16
            rows = slice(b * L, (b + 1) * L)
            s = ops.gemm_f32out(q[rows], k[rows], alpha=C ** -0.5)
            if Lp == L:
                p = ops.softmax_rows(s)
                vt = ops.gemm(pk.v[0], t[rows])              # V^T [C, L] = W_v x^T (bias deferred: rows of P sum to 1)
            else:
                ops.softmax_rows(s, out=p_pad[:, :L])
                ops.gemm(pk.v[0], t[rows], out=vt_pad[:, :L])
                p, vt = p_pad, vt_pad
            ops.gemm(p, vt, pk.v[1], out=a[rows])            # P V + b_v
        return ops.gemm(a, pk.o[0], pk.o[1], residual=x)


class AutoencoderKLDecoder(_VAEHalf):
    """Parameters under diffusers' names; ``decode`` / ``decode_latents`` run on the HIP kernels."""

    def __init__(self, config: Optional[VAEConfig] = None, ops=None, device: Optional[Union[str, torch.device]] = None):
        super().__init__(config, ops)
        cfg = self.config
        with (torch.device(device) if device is not None else torch.device("cpu")):
            self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)
            self.decoder = _Decoder(cfg)

    def _pack(self):
        d, cfg = self.decoder, self.config
        P = SimpleNamespace()
        P.pq = (self.post_quant_conv.weight.detach().float().reshape(cfg.latent_channels, cfg.latent_channels).contiguous(),
                self._f(self.post_quant_conv.bias))
        wi = d.conv_in.weight.detach().float().permute(0, 2, 3, 1).reshape(d.conv_in.weight.shape[0], -1)
        wpad = wi.new_zeros(wi.shape[0], 64)
        wpad[:, : wi.shape[1]] = wi                        # conv_in as a K = 64 GEMM over im2col patches
        P.conv_in = (self._w(wpad), self._f(d.conv_in.bias))
        a = d.mid_block.attentions[0]
        P.mid = SimpleNamespace(
            r0=self._pack_resnet(d.mid_block.resnets[0]), r1=self._pack_resnet(d.mid_block.resnets[1]),
            gn=(self._f(a.group_norm.weight), self._f(a.group_norm.bias)),
            q=(self._w(a.to_q.weight), self._f(a.to_q.bias)), k=(self._w(a.to_k.weight), self._f(a.to_k.bias)),
            v=(self._w(a.to_v.weight), self._f(a.to_v.bias)), o=(self._w(a.to_out[0].weight), self._f(a.to_out[0].bias)))
        P.up = [SimpleNamespace(res=[self._pack_resnet(r) for r in b.resnets],
                                up=None if b.upsamplers is None else self._conv_w(b.upsamplers[0].conv)) for b in d.up_blocks]
        P.norm_out = (self._f(d.conv_norm_out.weight), self._f(d.conv_norm_out.bias))
        P.conv_out = self._conv_w(d.conv_out, pad_out_to=4)
        self._packed = P
        return P

    @torch.no_grad()
    @on_model_device
    def decode(self, z: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """AutoencoderKL.decode(z).sample for z [B, 4, h, w] (any float dtype) -> fp32 image [B, 3, 8h, 8w];
        ``scale`` multiplies z first (decode_latents passes 1 / scaling_factor)."""
        ops, cfg = self.ops, self.config
        P = self._packed if self._packed is not None else self._pack()
        B, Cz, H, W = z.shape
        z = ops.channel_mix(z.to(device=self.device, dtype=torch.float32).contiguous(), P.pq[0], P.pq[1], scale)
        x = ops.gemm(ops.im2col_in(z.reshape(B, Cz, 1, H, W)), P.conv_in[0], P.conv_in[1])
        x = self._resnet(x, B, H, W, P.mid.r0)
        x = self._mid_attention(x, B, H, W, P.mid)
        x = self._resnet(x, B, H, W, P.mid.r1)
        for blk in P.up:
            for r in blk.res:
                x = self._resnet(x, B, H, W, r)
            if blk.up is not None:
                x, H, W = ops.conv3x3(x, B, H, W, blk.up[0], blk.up[1], up2x=True)
        x = ops.group_norm(x, B, H * W, P.norm_out[0], P.norm_out[1], cfg.norm_num_groups, 1e-6, True)
        x, _, _ = ops.conv3x3(x, B, H, W, P.conv_out[0], P.conv_out[1])
        img = ops.unpack_out(x, B, 4, 1, H, W, torch.float32)        # [B, 4, 1, H, W]; channel 3 is the padding
        return img[:, : cfg.out_channels, 0].contiguous()

    @torch.no_grad()
    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """pipeline.py:566-579: latents [b, 4, F, h, w] -> video [b, 3, F, 8h, 8w] float32."""
        b, c, f, h, w = latents.shape
        z = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        image = self.decode(z, scale=1.0 / self.config.scaling_factor)
        return image[None, :].reshape((b, f, -1) + image.shape[2:]).permute(0, 2, 1, 3, 4).float()


class AutoencoderKLEncoder(_VAEHalf):
    """The encoder half under diffusers' names (``encoder.*``, ``quant_conv.*``): ``encode`` returns the posterior moments,
    ``encode_latents`` is pipeline.py:556-560 (sample, times scaling_factor)."""

    def __init__(self, config: Optional[VAEConfig] = None, ops=None, device: Optional[Union[str, torch.device]] = None):
        super().__init__(config, ops)
        cfg = self.config
        with (torch.device(device) if device is not None else torch.device("cpu")):
            self.encoder = _Encoder(cfg)
            self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)

    def _pack(self):
        e, cfg = self.encoder, self.config
        P = SimpleNamespace()
        wi = e.conv_in.weight.detach().float().permute(0, 2, 3, 1).reshape(e.conv_in.weight.shape[0], -1)
        wpad = wi.new_zeros(wi.shape[0], 64)
        wpad[:, : wi.shape[1]] = wi                        # conv_in as a K = 64 GEMM over im2col patches of the RGB image
        P.conv_in = (self._w(wpad), self._f(e.conv_in.bias))
        P.down = []
        for b in e.down_blocks:
            dn = None
            if b.downsamplers is not None:                 # flipped filter for the flipped image (module docstring)
                conv = b.downsamplers[0].conv
                w = conv.weight.detach().float().flip(2, 3).permute(0, 2, 3, 1).reshape(conv.weight.shape[0], -1)
                dn = (self._w(w), self._f(conv.bias))
            P.down.append(SimpleNamespace(res=[self._pack_resnet(r) for r in b.resnets], down=dn))
        a = e.mid_block.attentions[0]
        P.mid = SimpleNamespace(
            r0=self._pack_resnet(e.mid_block.resnets[0]), r1=self._pack_resnet(e.mid_block.resnets[1]),
            gn=(self._f(a.group_norm.weight), self._f(a.group_norm.bias)),
            q=(self._w(a.to_q.weight), self._f(a.to_q.bias)), k=(self._w(a.to_k.weight), self._f(a.to_k.bias)),
            v=(self._w(a.to_v.weight), self._f(a.to_v.bias)), o=(self._w(a.to_out[0].weight), self._f(a.to_out[0].bias)))
        P.norm_out = (self._f(e.conv_norm_out.weight), self._f(e.conv_norm_out.bias))
        P.conv_out = self._conv_w(e.conv_out)
        c2 = 2 * cfg.latent_channels
        P.quant = (self.quant_conv.weight.detach().float().reshape(c2, c2).contiguous(), self._f(self.quant_conv.bias))
        self._packed = P
        return P

    def _downsample(self, x, B, H, W, pk):
        """Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) then conv3x3 stride 2 == flip(conv3x3_s2_p1(flip(x), flip(w)))."""
        if H % 2 or W % 2:
            raise ValueError(f"VAE encoder needs even feature maps at every level, got {H}x{W}")
        C = x.shape[1]
        xf = x.reshape(B, H, W, C).flip(1, 2).reshape(B * H * W, C).contiguous()
        y, Ho, Wo = self.ops.conv3x3(xf, B, H, W, pk[0], pk[1], stride=2)
        Co = y.shape[1]
        return y.reshape(B, Ho, Wo, Co).flip(1, 2).reshape(B * Ho * Wo, Co).contiguous(), Ho, Wo

    @torch.no_grad()
    @on_model_device
    def encode(self, images: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """AutoencoderKL.encode(x).latent_dist for x [B, 3, H, W] in [-1, 1]: (mean, logvar) fp32 [B, 4, H/8, W/8], logvar
        clamped to [-30, 20] as DiagonalGaussianDistribution does."""
        ops, cfg = self.ops, self.config
        P = self._packed if self._packed is not None else self._pack()
        B, Ci, H, W = images.shape
        levels = len(cfg.block_out_channels) - 1
        if Ci != cfg.out_channels or H % (1 << levels) or W % (1 << levels):
            raise ValueError(f"images must be [B, {cfg.out_channels}, H, W] with H, W multiples of {1 << levels}")
        img = images.to(device=self.device, dtype=torch.float32).contiguous()
        x = ops.gemm(ops.im2col_in(img.reshape(B, Ci, 1, H, W)), P.conv_in[0], P.conv_in[1])
        for blk in P.down:
            for r in blk.res:
                x = self._resnet(x, B, H, W, r)
            if blk.down is not None:
                x, H, W = self._downsample(x, B, H, W, blk.down)
        x = self._resnet(x, B, H, W, P.mid.r0)
        x = self._mid_attention(x, B, H, W, P.mid)
        x = self._resnet(x, B, H, W, P.mid.r1)
        x = ops.group_norm(x, B, H * W, P.norm_out[0], P.norm_out[1], cfg.norm_num_groups, 1e-6, True)
        x, _, _ = ops.conv3x3(x, B, H, W, P.conv_out[0], P.conv_out[1])
        c2 = 2 * cfg.latent_channels
        moments = ops.unpack_out(x, B, c2, 1, H, W, torch.float32)[:, :, 0].contiguous()
        moments = ops.channel_mix(moments, P.quant[0], P.quant[1], 1.0)             # quant_conv (1x1) in fp32
        mean, logvar = moments[:, : cfg.latent_channels], moments[:, cfg.latent_channels:]
        return mean.contiguous(), logvar.clamp(-30.0, 20.0).contiguous()

    @torch.no_grad()
    def encode_latents(self, images: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """pipeline.py:556-560: ``vae.encode(x).latent_dist.sample() * scaling_factor`` -> [B, 4, H/8, W/8] fp32."""
        mean, logvar = self.encode(images)
        noise = randn_like_reference(mean.shape, generator, mean.device, mean.dtype)
        return (mean + torch.exp(0.5 * logvar) * noise) * self.config.scaling_factor
