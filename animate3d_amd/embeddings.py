"""Input-independent positional tables of the path, computed once on the host in fp32.

* ``sine_pos_2d``      — reference SinePositionalEncoding2D(normalize=True)._forward,
                         animatediff/models/embeddings.py:59-96 (all-valid mask).
* ``sinusoidal_pos_1d``— diffusers SinusoidalPositionalEmbedding.pe, which the reference keeps in
                         the motion processors as ``time_pos_embed`` (attention_processor.py:497).
* ``get_camera``       — pipeline.py:127-190 (elevation 15 deg, azimuth sweep, unit-sphere normalise).
"""
from __future__ import annotations

import math

import torch


def sine_pos_2d(num_feats: int, h: int, w: int, temperature: float = 10000.0, scale: float = 2 * math.pi,
                eps: float = 1e-6) -> torch.Tensor:
    """-> fp32 [h*w, 2*num_feats] token-major table (channels: y-half then x-half, sin/cos interleaved)."""
    ys = torch.arange(1, h + 1, dtype=torch.float32)
    xs = torch.arange(1, w + 1, dtype=torch.float32)
    ys = ys / (float(h) + eps) * scale
    xs = xs / (float(w) + eps) * scale
    idx = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(idx, 2, rounding_mode="floor") / num_feats)

    def interleave(pos):                      # pos [n] -> [n, num_feats] (sin on even, cos on odd channels)
        a = pos[:, None] / dim_t[None, :]
        out = torch.empty_like(a)
        out[:, 0::2] = a[:, 0::2].sin()
        out[:, 1::2] = a[:, 1::2].cos()
        return out

    py = interleave(ys)[:, None, :].expand(h, w, num_feats)
    px = interleave(xs)[None, :, :].expand(h, w, num_feats)
    return torch.cat([py, px], dim=2).reshape(h * w, 2 * num_feats).contiguous()


def sinusoidal_pos_1d(embed_dim: int, max_seq_length: int) -> torch.Tensor:
    """-> fp32 [1, max_seq_length, embed_dim]."""
    position = torch.arange(max_seq_length, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, embed_dim, 2, dtype=torch.float32) * (-math.log(10000.0) / embed_dim))
    pe = torch.zeros(1, max_seq_length, embed_dim)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def get_camera(num_views: int, elevation: float = 15.0, azimuth_start: float = 0.0, azimuth_span: float = 360.0) -> torch.Tensor:
    """Camera conditioning of the pipeline: [num_views, 16] flattened, translation-normalised c2w."""
    el = math.radians(elevation)
    out = []
    for i in range(num_views):
        az = math.radians(azimuth_start + i * (azimuth_span / num_views))
        pos = torch.tensor([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)], dtype=torch.float32)
        fwd = torch.nn.functional.normalize(-pos, dim=0)
        right = torch.nn.functional.normalize(torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0])), dim=0)
        up = torch.nn.functional.normalize(torch.linalg.cross(right, fwd), dim=0)
        m = torch.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2] = right, up, -fwd
        m[:3, 3] = pos / (pos.norm() + 1e-8)
        out.append(m.flatten())
    return torch.stack(out, 0)
