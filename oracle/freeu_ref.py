"""FreeU on the CPU oracle (TEST INFRASTRUCTURE; see oracle/__init__.py).

``MVUNetMotionModel.enable_freeu`` (unet_motion_mv_model.py:562-585) only stores four factors on the up blocks; the arithmetic lives in
diffusers' up blocks (``apply_freeu`` / ``fourier_filter`` of diffusers.utils.torch_utils, v0.28.0), which are absent here like the rest of
diffusers: restated from their published form — **parity unpinned**.  Kept out of oracle/unet_ref.py so that the fingerprint of that file
(the committed full-size goldens carry it) does not move: ``freeu(ref, s1, s2, b1, b2)`` is a context manager that swaps the forward of
every up block of an ``MVUNetMotionModelRef`` for the FreeU-aware loop below.
"""
from __future__ import annotations

import contextlib

import torch


def fourier_filter(x_in: torch.Tensor, threshold: int, scale: float) -> torch.Tensor:
    """Scale the 2 threshold x 2 threshold centre of the shifted 2-D spectrum (the lowest frequencies) of every [H, W] plane."""
    x = x_in
    B, C, H, W = x.shape
    if (W & (W - 1)) != 0 or (H & (H - 1)) != 0:          # non-power-of-two planes are transformed in fp32
        x = x.to(dtype=torch.float32)
    x_freq = torch.fft.fftshift(torch.fft.fftn(x, dim=(-2, -1)), dim=(-2, -1))
    mask = torch.ones((B, C, H, W), device=x.device)
    crow, ccol = H // 2, W // 2
    mask[..., crow - threshold:crow + threshold, ccol - threshold:ccol + threshold] = scale
    x_freq = torch.fft.ifftshift(x_freq * mask, dim=(-2, -1))
    return torch.fft.ifftn(x_freq, dim=(-2, -1)).real.to(dtype=x_in.dtype)


def apply_freeu(resolution_idx: int, hidden_states, res_hidden_states, s1, s2, b1, b2):
    """Up block 0 / 1: the first half of the backbone channels x b1 / b2, the low frequencies of the skip features x s1 / s2 (FreeU,
    arXiv 2309.11497); later up blocks are untouched."""
    if resolution_idx in (0, 1):
        b, sc = (b1, s1) if resolution_idx == 0 else (b2, s2)
        half = hidden_states.shape[1] // 2
        hidden_states = hidden_states.clone()
        hidden_states[:, :half] = hidden_states[:, :half] * b
        res_hidden_states = fourier_filter(res_hidden_states, threshold=1, scale=sc)
    return hidden_states, res_hidden_states


def _up_forward(blk, idx, factors):
    """diffusers' CrossAttnUpBlockMotion / UpBlockMotion forward with FreeU on: oracle.unet_ref.UpBlockMotion.forward + apply_freeu in front of the cat."""
    def forward(x, res_tuple, temb, encoder_hidden_states, num_frames, upsample_size=None):
        for i, resnet in enumerate(blk.resnets):
            skip = res_tuple[-1]
            res_tuple = res_tuple[:-1]
            x, skip = apply_freeu(idx, x, skip, **factors)
            x = torch.cat([x, skip], dim=1)
            x = resnet(x, temb)
            if blk.has_cross_attention:
                x = blk.attentions[i](x, encoder_hidden_states)
            x = blk.motion_modules[i](x, num_frames)
        if blk.upsamplers is not None:
            x = blk.upsamplers[0](x, upsample_size)
        return x
    return forward


@contextlib.contextmanager
def freeu(ref, s1: float, s2: float, b1: float, b2: float):
    """``with freeu(ref, s1, s2, b1, b2): ref(**inputs)`` — the oracle with FreeU enabled (resolution_idx = the up block's position)."""
    factors = dict(s1=s1, s2=s2, b1=b1, b2=b2)
    saved = []
    try:
        for idx, blk in enumerate(ref.up_blocks):
            saved.append(blk.forward)
            blk.forward = _up_forward(blk, idx, factors)
        yield ref
    finally:
        for blk, f in zip(ref.up_blocks, saved):
            blk.forward = f
