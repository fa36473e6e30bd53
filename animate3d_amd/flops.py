"""Algorithmic work of one denoise step (SURVEY.md Appendix C conventions: 1 MAC = 2 FLOP; softmax /
normalisation / activation FLOPs excluded; text / IP K,V projections counted per image as the
reference executes them).  Used by bench.py to price rooflines and to scale the CPU-baseline sample."""
from __future__ import annotations

from .config import UNetConfig


def step_flops(cfg: UNetConfig, videos: int, views: int, frames: int, h: int, w: int, text_tokens: int = 77):
    """-> dict of FLOPs per component for one MVUNetMotionModel.forward on [videos, 4, frames, h, w]."""
    boc = cfg.block_out_channels
    nlev = len(boc)
    B2 = videos * frames
    G = (videos // views) * frames
    out = dict(conv=0.0, t2d_gemm=0.0, motion_gemm=0.0, mv_attn=0.0, i2v_attn=0.0, sp_attn=0.0, temporal_attn=0.0, cross_attn=0.0)

    def conv(cin, cout, hh, ww, k=3):
        out["conv"] += 2.0 * B2 * hh * ww * cin * cout * k * k

    def resnet(cin, cout, hh, ww):
        conv(cin, cout, hh, ww)
        conv(cout, cout, hh, ww)
        out["conv"] += 2.0 * B2 * cfg.block_out_channels[0] * 4 * cout          # time_emb_proj per image
        if cin != cout:
            conv(cin, cout, hh, ww, 1)

    def t2d(c, hh, ww):
        L = hh * ww
        tok = B2 * L
        n_mats = 22 if cfg.mvdream_image_attn else 20
        out["t2d_gemm"] += 2.0 * n_mats * c * c * tok
        out["t2d_gemm"] += B2 * 2.0 * (text_tokens + cfg.ip_num_tokens) * cfg.cross_attention_dim * c * 2
        S = views * L
        out["mv_attn"] += 4.0 * S * S * c * G
        if cfg.mvdream_image_attn:
            out["i2v_attn"] += 4.0 * S * S * c * G
        out["cross_attn"] += 4.0 * L * (text_tokens + cfg.ip_num_tokens) * c * B2

    def motion(c, hh, ww):
        L = hh * ww
        tok = B2 * L
        n_mats = 30 if cfg.motion_spatial_attn else 22
        out["motion_gemm"] += 2.0 * n_mats * c * c * tok
        S = views * L
        if cfg.motion_spatial_attn:
            out["sp_attn"] += 2 * 4.0 * S * S * c * G
        out["temporal_attn"] += 2 * 4.0 * frames * frames * c * videos * L

    conv(cfg.in_channels, boc[0], h, w)
    n = cfg.layers_per_block
    c_prev = boc[0]
    for i in range(nlev):
        hh, ww = h >> i, w >> i
        for j in range(n):
            resnet(c_prev if j == 0 else boc[i], boc[i], hh, ww)
            if cfg.down_has_attn[i]:
                t2d(boc[i], hh, ww)
            motion(boc[i], hh, ww)
        c_prev = boc[i]
        if i != nlev - 1:
            conv(boc[i], boc[i], hh >> 1, ww >> 1)
    hh, ww = h >> (nlev - 1), w >> (nlev - 1)
    c = boc[-1]
    resnet(c, c, hh, ww); t2d(c, hh, ww); motion(c, hh, ww); resnet(c, c, hh, ww)
    rev, rev_attn = list(reversed(boc)), list(reversed(cfg.down_has_attn))
    out_c = rev[0]
    for i in range(nlev):
        lev = nlev - 1 - i
        hh, ww = h >> lev, w >> lev
        prev_c, out_c = out_c, rev[i]
        in_c = rev[min(i + 1, nlev - 1)]
        for j in range(n + 1):
            skip_c = in_c if j == n else out_c
            res_in = prev_c if j == 0 else out_c
            resnet(res_in + skip_c, out_c, hh, ww)
            if rev_attn[i]:
                t2d(out_c, hh, ww)
            motion(out_c, hh, ww)
        if i != nlev - 1:
            conv(out_c, out_c, hh * 2, ww * 2)
    conv(boc[0], cfg.out_channels, h, w)
    out["total"] = sum(out.values())
    return out


def vae_decode_flops(h: int, w: int, block_out_channels=(128, 256, 512, 512), layers_per_block: int = 2,
                     latent_channels: int = 4, out_channels: int = 3) -> float:
    """Algorithmic FLOPs (1 MAC = 2 FLOP; norms / activations excluded) of AutoencoderKL.decode for ONE h x w latent
    (SD1.5 VAE decoder: conv_in, mid block with one single-head attention, four up blocks of three resnets, conv_out)."""
    conv = lambda px, cin, cout, k=9: 2.0 * px * k * cin * cout
    rev = list(reversed(block_out_channels))
    px = h * w
    c = rev[0]
    total = conv(px, latent_channels, c)
    total += 4 * conv(px, c, c)                                   # two mid resnets
    total += 4 * 2.0 * px * c * c + 2 * 2.0 * px * px * c         # q, k, v, out projections + QK^T and PV
    cin = c
    for i, cout in enumerate(rev):
        for j in range(layers_per_block + 1):
            ci = cin if j == 0 else cout
            total += conv(px, ci, cout) + conv(px, cout, cout) + (conv(px, ci, cout, 1) if ci != cout else 0.0)
        cin = cout
        if i != len(rev) - 1:
            px *= 4
            total += conv(px, cout, cout)                         # nearest-2x upsample + conv
    return total + conv(px, block_out_channels[0], out_channels)
