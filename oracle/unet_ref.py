"""Plain-PyTorch fp32 CPU restatement of Animate3D's MV-VDM UNet forward.

THIS FILE IS TEST INFRASTRUCTURE (the parity oracle and the CPU baseline).  It is
never imported by the product package ``animate3d_amd``.

Parity status
-------------
* The four attention processors and the 2-D sine positional encoding follow
  ``/root/reference/animatediff/models/attention_processor.py`` and
  ``embeddings.py``; they are PINNED by golden vectors generated from the
  reference's own code (``tests/golden/make_processor_goldens.py`` ->
  ``tests/golden/processors.npz``; checked by ``tests/test_oracle_golden.py``).
* The UNet glue follows ``animatediff/models/unet_motion_mv_model.py:633-867`` and is PINNED as well: the reference's own
  ``forward``, compiled from its syntax tree and run over this file's blocks, gives bit-identical outputs
  (``tests/golden/make_unet_forward_goldens.py`` -> ``tests/golden/unet_forward.npz``).
* Everything the reference imports from ``diffusers==0.28.0`` (ResnetBlock2D,
  Transformer2DModel, TransformerTemporalModel, BasicTransformerBlock, GEGLU
  feed-forward, Attention, Timesteps, TimestepEmbedding, ImageProjection,
  Down/Upsample2D, AlphaBlender, SinusoidalPositionalEmbedding, the *Motion
  blocks) is restated from that library's published semantics (SURVEY.md
  Appendix A).  diffusers is NOT installed here and the reference holds no tests
  or golden tensors at that boundary: **parity unpinned** for those pieces.

Layout is the reference's: NCHW images ``[(b n f), C, h, w]``, tokens
``[(b n f), h*w, C]`` on the 2-D side and ``[(b n h w), f, C]`` in the motion
modules.  Module / parameter names follow diffusers so a reference state-dict
loads key-for-key.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# config
# --------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    """Constructor constants of MVUNetMotionModel (unet_motion_mv_model.py:67-102) with the
    values the released checkpoints use."""

    sample_size: Optional[int] = 32
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    cross_attention_dim: int = 768
    num_attention_heads: int = 8
    motion_num_attention_heads: int = 8
    motion_max_seq_length: int = 32
    camera_embedding_dim: Optional[int] = 16
    ip_image_embed_dim: Optional[int] = 1024   # IP-Adapter sd15 ImageProjection input
    ip_num_tokens: int = 4
    ip_scale: float = 1.0
    # processor switches = configs/inference/inference.yaml:9-24 (released values)
    mvdream_image_attn: bool = True
    motion_spatial_attn: bool = True
    motion_use_spatial_encoding: bool = True
    motion_use_alpha_blender: bool = True
    # switches the released configs leave off (motion_module_attn_cfg.image_attn / spatial_attn.attn_cfg.*, inference.yaml:12-24)
    motion_image_attn: bool = False
    motion_use_camera_encoding: bool = False
    motion_spatial_encoding_type: str = "sinusoid"       # or "learnable"
    motion_camera_encoding_type: str = "sinusoid"        # or "learnable"
    encoder_hid_dim_type: Optional[str] = "ip_image_proj"

    def to_dict(self):
        return dict(self.__dict__)


# --------------------------------------------------------------------------------------
# embeddings
# --------------------------------------------------------------------------------------
def timestep_sinusoid(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)
    (call site unet_motion_mv_model.py:133,723).  Returns fp32 [N, dim] = [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    freqs = torch.exp(exponent)
    arg = timesteps[:, None].float() * freqs[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


class TimestepEmbedding(nn.Module):
    """diffusers TimestepEmbedding: linear_1 -> SiLU -> linear_2."""

    def __init__(self, in_dim: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ImageProjection(nn.Module):
    """diffusers ImageProjection (IP-Adapter sd15): Linear(1024 -> T*768), reshape, LayerNorm."""

    def __init__(self, image_embed_dim: int, cross_attention_dim: int, num_image_text_embeds: int):
        super().__init__()
        self.num_image_text_embeds = num_image_text_embeds
        self.image_embeds = nn.Linear(image_embed_dim, num_image_text_embeds * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)

    def forward(self, image_embeds):
        b = image_embeds.shape[0]
        x = self.image_embeds(image_embeds).reshape(b, self.num_image_text_embeds, -1)
        return self.norm(x)


class MultiIPAdapterImageProjection(nn.Module):
    """diffusers MultiIPAdapterImageProjection with one adapter; a bare tensor [V, 1024]
    takes the single-image path -> list of one [V, T, 768] tensor."""

    def __init__(self, layers):
        super().__init__()
        self.image_projection_layers = nn.ModuleList(layers)

    def forward(self, image_embeds):
        if not isinstance(image_embeds, (list, tuple)):
            image_embeds = [image_embeds]
        return [layer(e.reshape(e.shape[0], -1)) for e, layer in zip(image_embeds, self.image_projection_layers)]


def sine_pos_2d(num_feats: int, h: int, w: int, temperature: float = 10000.0,
                scale: float = 2 * math.pi, eps: float = 1e-6) -> torch.Tensor:
    """SinePositionalEncoding2D(num_feats, normalize=True)._forward on an all-valid mask
    (embeddings.py:59-96).  Returns fp32 [2*num_feats, h, w] (y half first, then x half)."""
    y_embed = torch.arange(1, h + 1, dtype=torch.float32)[:, None].expand(h, w)
    x_embed = torch.arange(1, w + 1, dtype=torch.float32)[None, :].expand(h, w)
    y_embed = y_embed / (y_embed[-1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_feats)
    pos_x = x_embed[:, :, None] / dim_t
    pos_y = y_embed[:, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).reshape(h, w, -1)
    pos_y = torch.stack((pos_y[:, :, 0::2].sin(), pos_y[:, :, 1::2].cos()), dim=3).reshape(h, w, -1)
    return torch.cat((pos_y, pos_x), dim=2).permute(2, 0, 1).contiguous()


def sinusoidal_pos_1d(embed_dim: int, max_seq_length: int) -> torch.Tensor:
    """diffusers SinusoidalPositionalEmbedding.pe  -> [1, max_seq_length, embed_dim]
    (instantiated by the reference at attention_processor.py:497)."""
    position = torch.arange(max_seq_length, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, embed_dim, 2, dtype=torch.float32) * (-math.log(10000.0) / embed_dim))
    pe = torch.zeros(1, max_seq_length, embed_dim)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


class TimePosEmbed(nn.Module):
    """Holds the ``pe`` buffer under the reference's key ``...processor.time_pos_embed.pe``."""

    def __init__(self, embed_dim: int, max_seq_length: int):
        super().__init__()
        self.register_buffer("pe", sinusoidal_pos_1d(embed_dim, max_seq_length))

    def forward(self, x):
        return x + self.pe[:, : x.shape[1]].to(x.dtype)


class AlphaBlender(nn.Module):
    """diffusers AlphaBlender(alpha, merge_strategy='learned'): sigmoid(mix_factor) weights
    the FIRST argument (call attention_processor.py:709)."""

    def __init__(self, alpha: float = 0.0):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([alpha], dtype=torch.float32))

    def forward(self, x_spatial, x_temporal):
        a = torch.sigmoid(self.mix_factor).to(x_spatial.dtype)
        return a * x_spatial + (1.0 - a) * x_temporal


class SoftmaxAlphaBlender(nn.Module):
    """attention_processor.py:727-744: three-way blend, softmax over ``mix_factor``; argument order (spatial, temporal, image)."""

    def __init__(self, alphas=(0.0, 0.0, 0.0)):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor(list(alphas), dtype=torch.float32))

    def forward(self, x_spatial, x_temporal, x_image):
        a = torch.softmax(self.mix_factor, dim=0).to(x_spatial.dtype)
        return x_spatial * a[0] + x_temporal * a[1] + x_image * a[2]


class LearnedPositionalEncoding2D(nn.Module):
    """embeddings.py:99-157: channels [0, num_feats) = col_embed(x), [num_feats, 2 num_feats) = row_embed(y)."""

    def __init__(self, num_feats: int, row_num_embed: int = 50, col_num_embed: int = 50):
        super().__init__()
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def table(self, h: int, w: int) -> torch.Tensor:
        """[h, w, 2 num_feats]"""
        xe = self.col_embed.weight[:w][None, :, :].expand(h, w, -1)
        ye = self.row_embed.weight[:h][:, None, :].expand(h, w, -1)
        return torch.cat([xe, ye], dim=-1)


class LabelEmbedding(nn.Module):
    """diffusers LabelEmbedding(num_classes, hidden_size, dropout_prob = 0): a plain table (attention_processor.py:508)."""

    def __init__(self, num_classes: int, hidden_size: int):
        super().__init__()
        self.embedding_table = nn.Embedding(num_classes, hidden_size)


# --------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------
def _sdpa(q, k, v, heads: int):
    """softmax(q k^T / sqrt(d)) v on [B, L, H*D] tensors -> [B, Lq, H*D]  (what
    xformers.ops.memory_efficient_attention computes at attention_processor.py:103 etc.).  The score matrix is
    materialised per batch chunk (<= 2^28 scores at a time) so that full-size configurations fit in host memory; the
    arithmetic per batch element does not depend on the chunking."""
    b, lq, c = q.shape
    d = c // heads
    lk = k.shape[1]
    step = max(1, (1 << 28) // max(1, heads * lq * lk))
    outs = []
    for i in range(0, b, step):
        qh = q[i:i + step].reshape(-1, lq, heads, d).transpose(1, 2)
        kh = k[i:i + step].reshape(-1, lk, heads, d).transpose(1, 2)
        vh = v[i:i + step].reshape(-1, lk, heads, d).transpose(1, 2)
        s_ = torch.matmul(qh, kh.transpose(-1, -2)) * (d ** -0.5)
        p = torch.softmax(s_, dim=-1)
        outs.append(torch.matmul(p, vh).transpose(1, 2).reshape(-1, lq, c))
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


def reference_attention_mask_trace(mask_shape, V, n, F, heads, tokens_per_image):
    """What the reference does with ``attention_mask`` [B, K] at every Transformer2D ``attn1`` it reaches — the shape walk behind the
    product's error for this argument (animate3d_amd/unet.py).

    * unet_motion_mv_model.py:700-703: ``(1 - mask) * -10000`` and ``unsqueeze(1)`` -> [B, 1, K]; :778, 806, 815, 841 hand it to every
      down / mid / up block, diffusers passes it to ``BasicTransformerBlock.attn1`` (the motion modules receive no mask).
    * attention_processor.py:340 regroups ``(b n f) l c -> (b f) (n l) c``: batch_size = b * F, key_tokens = query_tokens = n * l; :361
      calls ``attn.prepare_attention_mask(mask, key_tokens, batch_size)`` — diffusers 0.28 (pinned by the reference's requirements;
      not under /root/reference): if the mask's last dimension differs from key_tokens it is PADDED BY key_tokens (``F.pad(mask, (0,
      target_length))``, not to key_tokens), and it is ``repeat_interleave``d over the heads only when its batch is smaller than
      batch_size * heads; :370 expands the singleton query dimension; :405, 416 pass it to xformers as ``attn_bias``, which requires
      exactly [batch_size * heads, query_tokens, key_tokens].

    ``tokens_per_image``: l = h * w at every resolution that has a Transformer2D (level 0 .. the mid block).  Returns one record per
    resolution: (l, bias shape the reference builds, shape xformers requires, consistent?).  A UNet with attention at two resolutions can
    never be consistent at both (K cannot equal n * l for two different l), i.e. the reference raises inside xformers for ANY mask."""
    B, K = mask_shape
    b = V // n
    out = []
    for l in tokens_per_image:
        batch, key_tokens = b * F, n * l
        k_eff = K if K == key_tokens else K + key_tokens
        b_eff = B * heads if B < batch * heads else B
        built, need = (b_eff, key_tokens, k_eff), (batch * heads, key_tokens, key_tokens)
        out.append((l, built, need, built == need))
    return out


class Attention(nn.Module):
    """The subset of diffusers.models.attention_processor.Attention that the reference's
    processors touch (SURVEY.md §8c list)."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8,
                 dim_head: int = 64, bias: bool = False, out_bias: bool = True):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = False
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])
        self.processor = None

    # -- helpers used by the reference processors (needed when this class is the stand-in
    #    handed to the reference's own code by tests/golden/make_processor_goldens.py) --
    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        assert attention_mask is None, "callers always pass attention_mask=None (SURVEY.md §3.2)"
        return None

    def head_to_batch_dim(self, t):
        b, l, c = t.shape
        return t.reshape(b, l, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, l, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, l, d = t.shape
        b = bh // self.heads
        return t.reshape(b, self.heads, l, d).permute(0, 2, 1, 3).reshape(b, l, d * self.heads)

    def get_attention_scores(self, query, key, attention_mask=None):
        s = torch.baddbmm(torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype),
                          query, key.transpose(-1, -2), beta=0, alpha=self.scale)
        return s.softmax(dim=-1)

    def set_processor(self, processor):
        self.processor = processor

    def get_processor(self, return_deprecated_lora=False):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class MVDreamProc(nn.Module):
    """Restates MVDreamXFormersAttnProcessor.__call__ (attention_processor.py:39-126):
    multi-view self-attention over the n*l tokens of each (b, f) group."""

    def __init__(self, num_views: int, num_frames: int):
        super().__init__()
        self.num_views, self.num_frames = num_views, num_frames

    def forward(self, attn: Attention, x, encoder_hidden_states=None, attention_mask=None, **kw):
        n, f = self.num_views, self.num_frames
        cross = encoder_hidden_states is not None
        if not cross:
            x = _bnf_to_bf_nl(x, n, f)                       # :54
        ctx = x if not cross else encoder_hidden_states
        o = _sdpa(attn.to_q(x), attn.to_k(ctx), attn.to_v(ctx), attn.heads)   # :89-105
        o = attn.to_out[0](o)                                 # :110
        if not cross:
            o = _bf_nl_to_bnf(o, n, f)                        # :124
        return o


class MVDreamI2VProc(nn.Module):
    """Restates MVDreamI2VXFormersAttnProcessor (attention_processor.py:302-445)."""

    def __init__(self, hidden_size: int, num_views: int, num_frames: int):
        super().__init__()
        self.num_views, self.num_frames = num_views, num_frames
        self.to_q_i2v = nn.Linear(hidden_size, hidden_size, bias=False)   # :322
        self.to_out_i2v = nn.Linear(hidden_size, hidden_size, bias=True)  # :323

    def forward(self, attn: Attention, x, encoder_hidden_states=None, attention_mask=None, **kw):
        n, f = self.num_views, self.num_frames
        assert encoder_hidden_states is None
        x = _bnf_to_bf_nl(x, n, f)                                        # :340  -> [(b f), n*l, c]
        q, k, v = attn.to_q(x), attn.to_k(x), attn.to_v(x)                # :375-383
        bf, s, c = k.shape
        # first-frame K/V of every b, broadcast to all f (:389-397)
        k0 = k.reshape(bf // f, f, s, c)[:, 0:1].expand(-1, f, -1, -1).reshape(bf, s, c)
        v0 = v.reshape(bf // f, f, s, c)[:, 0:1].expand(-1, f, -1, -1).reshape(bf, s, c)
        main = _sdpa(q, k, v, attn.heads)                                 # :405-409
        i2v = _sdpa(self.to_q_i2v(x), k0, v0, attn.heads)                 # :413-420
        i2v = self.to_out_i2v(i2v)                                        # :423
        o = attn.to_out[0](main + i2v)                                    # :426-429
        return _bf_nl_to_bnf(o, n, f)                                     # :443


class IPAdapterProc(nn.Module):
    """Restates IPAdapterXFormersAttnProcessor (attention_processor.py:129-298), mask-free path."""

    def __init__(self, hidden_size: int, cross_attention_dim: int, num_tokens=(4,), scale=1.0):
        super().__init__()
        self.hidden_size, self.cross_attention_dim = hidden_size, cross_attention_dim
        self.num_tokens = list(num_tokens)
        self.scale = [scale] * len(self.num_tokens)
        self.to_k_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in self.num_tokens])
        self.to_v_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in self.num_tokens])

    def forward(self, attn: Attention, x, encoder_hidden_states=None, attention_mask=None, **kw):
        text, ip_list = encoder_hidden_states                              # :182-184 (tuple form)
        q = attn.to_q(x)                                                   # :214
        o = _sdpa(q, attn.to_k(text), attn.to_v(text), attn.heads)         # :221-237
        for ip, scale, to_k_ip, to_v_ip in zip(ip_list, self.scale, self.to_k_ip, self.to_v_ip):
            o = o + scale * _sdpa(q, to_k_ip(ip), to_v_ip(ip), attn.heads)  # :254-283
        return attn.to_out[0](o)                                           # :286


class SpatioTemporalProc(nn.Module):
    """Restates SpatioTemporalI2VXFormersAttnProcessor (attention_processor.py:448-723): temporal attention, optional
    multi-view spatial attention (2-D positional encoding sinusoid / learnable, optional per-view camera encoding sinusoid /
    learnable), optional first-frame image attention, merged by sum, AlphaBlender (two branches) or SoftmaxAlphaBlender
    (three).  Input is ``[(b n h w), f, c]``."""

    def __init__(self, hidden_size: int, feature_size, num_views: int, num_frames: int,
                 spatial_attn: bool = True, use_spatial_encoding: bool = True, use_alpha_blender: bool = True,
                 max_seq_length: int = 32, image_attn: bool = False, use_camera_encoding: bool = False,
                 spatial_encoding_type: str = "sinusoid", camera_encoding_type: str = "sinusoid",
                 embed_size: Optional[int] = None):
        super().__init__()
        self.hidden_size = hidden_size
        self.feature_hw = (feature_size, feature_size) if isinstance(feature_size, int) else tuple(feature_size)
        self.num_views, self.num_frames = num_views, num_frames
        self.use_spatial_attn = spatial_attn
        self.use_spatial_encoding = use_spatial_encoding
        self.use_camera_encoding = use_camera_encoding
        self.spatial_encoding_type, self.camera_encoding_type = spatial_encoding_type, camera_encoding_type
        self.use_image_attn = image_attn
        self.use_alpha_blender = use_alpha_blender
        if spatial_attn:
            self.to_q_sp = nn.Linear(hidden_size, hidden_size, bias=False)     # :490-493
            self.to_k_sp = nn.Linear(hidden_size, hidden_size, bias=False)
            self.to_v_sp = nn.Linear(hidden_size, hidden_size, bias=False)
            self.to_out_sp = nn.Linear(hidden_size, hidden_size, bias=True)
            if use_spatial_encoding:
                self.time_pos_embed = TimePosEmbed(hidden_size, max_seq_length)  # :497
                if spatial_encoding_type == "learnable":                          # :502-503 (table size = constructor feature_size)
                    es = embed_size if embed_size is not None else max(self.feature_hw)
                    self.spatial_pos_embed = LearnedPositionalEncoding2D(hidden_size // 2, es, es)
                elif spatial_encoding_type != "sinusoid":
                    raise ValueError(f"Spatial encoding type {spatial_encoding_type} is not supported yet!")
            if use_camera_encoding:
                self.time_pos_embed = TimePosEmbed(hidden_size, max_seq_length)  # :508
                if camera_encoding_type == "learnable":
                    self.camera_embed = LabelEmbedding(num_views, hidden_size)   # :510
                elif camera_encoding_type == "sinusoid":
                    self.camera_embed = TimePosEmbed(hidden_size, num_views)     # :512
        if image_attn:                                                            # :514-518
            self.to_q_i2v = nn.Linear(hidden_size, hidden_size, bias=False)
            self.to_k_i2v = nn.Linear(hidden_size, hidden_size, bias=False)
            self.to_v_i2v = nn.Linear(hidden_size, hidden_size, bias=False)
            self.to_out_i2v = nn.Linear(hidden_size, hidden_size, bias=True)
        num_attn = 1 + int(spatial_attn) + int(image_attn)
        if not use_alpha_blender:                                                 # :527-536: zero-initialised extra branches
            for m in ([self.to_out_sp] if spatial_attn else []) + ([self.to_out_i2v] if image_attn else []):
                nn.init.zeros_(m.weight)
                nn.init.zeros_(m.bias)
        elif num_attn == 2:
            self.alpha_blender = AlphaBlender(0.0)                                # :537
        elif num_attn == 3:
            self.alpha_blender = SoftmaxAlphaBlender((0.0, 0.0, 0.0))             # :539

    def forward(self, attn: Attention, x, encoder_hidden_states=None, attention_mask=None, **kw):
        n, f = self.num_views, self.num_frames
        fh, fw = self.feature_hw
        assert encoder_hidden_states is None
        c = x.shape[-1]
        if self.use_spatial_attn:
            s = n * fh * fw
            bl, ff, _ = x.shape
            assert ff == f and bl % s == 0
            b = bl // s
            sp = x.reshape(b, s, f, c).permute(0, 2, 1, 3).reshape(b * f, s, c)       # :557
            if self.use_spatial_encoding:                                             # :559-563
                if self.spatial_encoding_type == "learnable":
                    pe = self.spatial_pos_embed.table(fh, fw).to(x.dtype).reshape(fh * fw, c)
                else:
                    pe = sine_pos_2d(c // 2, fh, fw).to(x.dtype).permute(1, 2, 0).reshape(fh * fw, c)
                sp = sp + pe.reshape(1, 1, fh * fw, c).expand(1, n, -1, -1).reshape(1, s, c)
            if self.use_camera_encoding:                                              # :565-575: one vector per view
                cam = self.camera_embed.embedding_table.weight[:n] if self.camera_encoding_type == "learnable" \
                    else self.camera_embed.pe[0, :n]
                sp = sp + cam.to(x.dtype).reshape(1, n, 1, c).expand(1, n, fh * fw, c).reshape(1, s, c)
        if self.use_image_attn:                                                       # :578-580 (before the time encoding)
            li = fh * fw
            img = x.reshape(-1, li, f, c).permute(0, 2, 1, 3)                         # [(b n), f, l, c]
        if self.use_spatial_attn and (self.use_spatial_encoding or self.use_camera_encoding):
            x = self.time_pos_embed(x)                                                # :583-584
        # temporal (AnimateDiff) branch :619-641, unfused softmax in the reference
        t = _sdpa(attn.to_q(x), attn.to_k(x), attn.to_v(x), attn.heads)
        t = attn.to_out[0](t)
        so = io = None
        if self.use_spatial_attn:
            so = _sdpa(self.to_q_sp(sp), self.to_k_sp(sp), self.to_v_sp(sp), attn.heads)  # :645-660
            so = self.to_out_sp(so)                                                       # :666
            so = so.reshape(b, f, s, c).permute(0, 2, 1, 3).reshape(b * s, f, c)          # :669
        if self.use_image_attn:                                                       # :672-698: K/V of frame 0, per view
            v_, _, li, _ = img.shape
            q_i = self.to_q_i2v(img.reshape(v_ * f, li, c))
            first = img[:, 0]                                                         # [(b n), l, c]
            k_i = self.to_k_i2v(first)[:, None].expand(v_, f, li, c).reshape(v_ * f, li, c)
            v_i = self.to_v_i2v(first)[:, None].expand(v_, f, li, c).reshape(v_ * f, li, c)
            io = self.to_out_i2v(_sdpa(q_i, k_i, v_i, attn.heads))
            io = io.reshape(v_, f, li, c).permute(0, 2, 1, 3).reshape(v_ * li, f, c)  # :698
        if not self.use_alpha_blender:                                                # :700-706
            out = t
            if so is not None:
                out = out + so
            if io is not None:
                out = out + io
            return out
        if so is not None and io is None:
            return self.alpha_blender(so, t)                                          # :709
        if io is not None and so is None:
            return self.alpha_blender(io, t)                                          # :711
        if so is not None and io is not None:
            return self.alpha_blender(so, t, io)                                      # :713
        return t


def _bnf_to_bf_nl(x, n, f):
    """einops '(b n f) l c -> (b f) (n l) c' (attention_processor.py:54,340)."""
    bnf, l, c = x.shape
    b = bnf // (n * f)
    return x.reshape(b, n, f, l, c).permute(0, 2, 1, 3, 4).reshape(b * f, n * l, c)


def _bf_nl_to_bnf(x, n, f):
    """einops '(b f) (n l) c -> (b n f) l c' (attention_processor.py:124,443)."""
    bf, nl, c = x.shape
    b, l = bf // f, nl // n
    return x.reshape(b, f, n, l, c).permute(0, 2, 1, 3, 4).reshape(b * n * f, l, c)


# --------------------------------------------------------------------------------------
# diffusers 0.28.0 blocks (parity unpinned)
# --------------------------------------------------------------------------------------
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """layer_norm variant: x += attn1(LN1 x); x += attn2(LN2 x, ctx); x += FF(LN3 x)."""

    def __init__(self, dim, heads, head_dim, cross_attention_dim=None, double_self_attention=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, heads, head_dim, bias=False)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, None if double_self_attention else cross_attention_dim, heads, head_dim, bias=False)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)
        self.double_self_attention = double_self_attention
        # diffusers gives a motion module's block a sinusoidal ``pos_embed`` (added to the LayerNorm output before attn1 AND attn2);
        # inference.py:176-192 sets it to None when the spatial branch carries an encoding (the processor then adds the temporal PE
        # itself).  A callable here (set by _install_processors for the other switch sets) stands for the kept module.
        self.pos_embed = None

    def forward(self, x, encoder_hidden_states=None):
        pe = self.pos_embed if self.pos_embed is not None else (lambda t: t)
        x = x + self.attn1(pe(self.norm1(x)))
        ctx = None if self.double_self_attention else encoder_hidden_states
        x = x + self.attn2(pe(self.norm2(x)), encoder_hidden_states=ctx)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    def __init__(self, heads, head_dim, in_channels, cross_attention_dim, groups=32):
        super().__init__()
        inner = heads * head_dim
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, head_dim, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, x, encoder_hidden_states):
        b, c, h, w = x.shape
        res = x
        x = self.proj_in(self.norm(x))
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states)
        x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2)
        return self.proj_out(x) + res


class TransformerTemporalModel(nn.Module):
    """AnimateDiff motion module: 3-D GroupNorm over (C/32, F, h, w)."""

    def __init__(self, heads, head_dim, in_channels, groups=32):
        super().__init__()
        inner = heads * head_dim
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, head_dim, None, double_self_attention=True)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, num_frames):
        bf, c, h, w = x.shape
        b = bf // num_frames
        res = x
        x = x.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x = self.norm(x)
        x = x.permute(0, 3, 4, 2, 1).reshape(b * h * w, num_frames, c)
        x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x)
        x = self.proj_out(x)
        x = x.reshape(b, h, w, num_frames, c).permute(0, 3, 4, 1, 2).reshape(bf, c, h, w)
        return x + res


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


class DownBlockMotion(nn.Module):
    """CrossAttnDownBlockMotion (has_attn) / DownBlockMotion."""

    def __init__(self, cfg: UNetConfig, in_c, out_c, temb_c, has_attn, add_downsample):
        super().__init__()
        self.has_cross_attention = has_attn
        n = cfg.layers_per_block
        self.resnets = nn.ModuleList([ResnetBlock2D(in_c if i == 0 else out_c, out_c, temb_c, cfg.norm_num_groups, cfg.norm_eps) for i in range(n)])
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cfg.num_attention_heads, out_c // cfg.num_attention_heads, out_c, cfg.cross_attention_dim, cfg.norm_num_groups) for _ in range(n)])
        self.motion_modules = nn.ModuleList([TransformerTemporalModel(cfg.motion_num_attention_heads, out_c // cfg.motion_num_attention_heads, out_c, cfg.norm_num_groups) for _ in range(n)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_c)]) if add_downsample else None

    def forward(self, x, temb, encoder_hidden_states, num_frames):
        outs = ()
        for i, resnet in enumerate(self.resnets):
            x = resnet(x, temb)
            if self.has_cross_attention:
                x = self.attentions[i](x, encoder_hidden_states)
            x = self.motion_modules[i](x, num_frames)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class MidBlockMotion(nn.Module):
    """UNetMidBlockCrossAttnMotion: resnet0 -> T2D -> motion -> resnet1."""

    def __init__(self, cfg: UNetConfig, c, temb_c):
        super().__init__()
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_c, cfg.norm_num_groups, cfg.norm_eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(cfg.num_attention_heads, c // cfg.num_attention_heads, c, cfg.cross_attention_dim, cfg.norm_num_groups)])
        self.motion_modules = nn.ModuleList([TransformerTemporalModel(cfg.motion_num_attention_heads, c // cfg.motion_num_attention_heads, c, cfg.norm_num_groups)])

    def forward(self, x, temb, encoder_hidden_states, num_frames):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, encoder_hidden_states)
        x = self.motion_modules[0](x, num_frames)
        return self.resnets[1](x, temb)


class UpBlockMotion(nn.Module):
    """CrossAttnUpBlockMotion (has_attn) / UpBlockMotion."""

    def __init__(self, cfg: UNetConfig, in_c, out_c, prev_c, temb_c, has_attn, add_upsample):
        super().__init__()
        self.has_cross_attention = has_attn
        n = cfg.layers_per_block + 1
        resnets = []
        for i in range(n):
            skip_c = in_c if i == n - 1 else out_c
            res_in = prev_c if i == 0 else out_c
            resnets.append(ResnetBlock2D(res_in + skip_c, out_c, temb_c, cfg.norm_num_groups, cfg.norm_eps))
        self.resnets = nn.ModuleList(resnets)
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cfg.num_attention_heads, out_c // cfg.num_attention_heads, out_c, cfg.cross_attention_dim, cfg.norm_num_groups) for _ in range(n)])
        self.motion_modules = nn.ModuleList([TransformerTemporalModel(cfg.motion_num_attention_heads, out_c // cfg.motion_num_attention_heads, out_c, cfg.norm_num_groups) for _ in range(n)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_c)]) if add_upsample else None

    def forward(self, x, res_tuple, temb, encoder_hidden_states, num_frames, upsample_size=None):
        for i, resnet in enumerate(self.resnets):
            skip = res_tuple[-1]
            res_tuple = res_tuple[:-1]
            x = torch.cat([x, skip], dim=1)
            x = resnet(x, temb)
            if self.has_cross_attention:
                x = self.attentions[i](x, encoder_hidden_states)
            x = self.motion_modules[i](x, num_frames)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, upsample_size)
        return x


# --------------------------------------------------------------------------------------
# the UNet
# --------------------------------------------------------------------------------------
class MVUNetMotionModelRef(nn.Module):
    """fp32 CPU restatement of MVUNetMotionModel (unet_motion_mv_model.py:55-867) with the
    processors of inference.py:90-192 installed for ``num_views`` / ``num_frames`` /
    latent ``(h, w)`` (feature sizes derived from the call shape)."""

    def __init__(self, cfg: UNetConfig, num_views: int, num_frames: int, latent_hw: Tuple[int, int]):
        super().__init__()
        self.cfg = cfg
        self.config = SimpleNamespace(**cfg.to_dict())
        self.num_views, self.num_frames = num_views, num_frames
        boc = cfg.block_out_channels
        temb_c = boc[0] * 4
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb_c)
        if cfg.camera_embedding_dim is not None:
            self.camera_embedding = TimestepEmbedding(cfg.camera_embedding_dim, temb_c)
        self.encoder_hid_proj = None
        if cfg.encoder_hid_dim_type == "ip_image_proj":
            self.encoder_hid_proj = MultiIPAdapterImageProjection(
                [ImageProjection(cfg.ip_image_embed_dim, cfg.cross_attention_dim, cfg.ip_num_tokens)])
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i in range(len(boc)):
            in_c, out_c = out_c, boc[i]
            self.down_blocks.append(DownBlockMotion(cfg, in_c, out_c, temb_c, cfg.down_has_attn[i], i != len(boc) - 1))
        self.mid_block = MidBlockMotion(cfg, boc[-1], temb_c)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        rev_attn = list(reversed(cfg.down_has_attn))
        out_c = rev[0]
        for i in range(len(boc)):
            prev_c, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(UpBlockMotion(cfg, in_c, out_c, prev_c, temb_c, rev_attn[i], i != len(boc) - 1))
        self.num_upsamplers = len(boc) - 1
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, boc[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)
        self._install_processors(latent_hw)

    # inference.py:90-192 restated; keyed on layer role, not class identity
    def _install_processors(self, latent_hw):
        cfg, n, f = self.cfg, self.num_views, self.num_frames
        nlev = len(cfg.block_out_channels)
        sizes = [tuple(latent_hw)]            # feature map per level: each Downsample2D (3x3, stride 2, pad 1) gives ceil(h / 2)
        for _ in range(nlev - 1):
            sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))

        def motion(c, hw, es):
            return SpatioTemporalProc(c, hw, n, f, cfg.motion_spatial_attn, cfg.motion_use_spatial_encoding,
                                      cfg.motion_use_alpha_blender, cfg.motion_max_seq_length, cfg.motion_image_attn,
                                      cfg.motion_use_camera_encoding, cfg.motion_spatial_encoding_type,
                                      cfg.motion_camera_encoding_type, embed_size=es)

        def t2d(tr, c):
            blk = tr.transformer_blocks[0]
            if cfg.mvdream_image_attn:
                p = MVDreamI2VProc(c, n, f)
                with torch.no_grad():                      # inference.py:161-165
                    p.to_q_i2v.weight.copy_(blk.attn1.to_q.weight)
                    p.to_out_i2v.weight.zero_()
                    p.to_out_i2v.bias.zero_()
            else:
                p = MVDreamProc(n, f)
            blk.attn1.set_processor(p)
            blk.attn2.set_processor(IPAdapterProc(c, cfg.cross_attention_dim, (cfg.ip_num_tokens,), cfg.ip_scale))

        def mm(m, c, hw, lvl):
            blk = m.transformer_blocks[0]
            es = max(max(hw), (cfg.sample_size or max(hw)) >> lvl)                    # learnable-PE table rows (inference.py:93-105)
            blk.attn1.set_processor(motion(c, hw, es))
            blk.attn2.set_processor(motion(c, hw, es))
            if not (cfg.motion_spatial_attn and (cfg.motion_use_spatial_encoding or cfg.motion_use_camera_encoding)):   # inference.py:176-178: pos_embed is kept
                with torch.device("cpu"):                  # a constant captured by the closure, not a buffer: must be real under build_dense's meta construction
                    table = sinusoidal_pos_1d(c, cfg.motion_max_seq_length)
                blk.pos_embed = lambda t, table=table: t + table[:, : t.shape[1]].to(t)

        for i, blk in enumerate(self.down_blocks):
            c = cfg.block_out_channels[i]
            for j in range(len(blk.resnets)):
                if blk.has_cross_attention:
                    t2d(blk.attentions[j], c)
                mm(blk.motion_modules[j], c, sizes[i], i)
        c = cfg.block_out_channels[-1]
        t2d(self.mid_block.attentions[0], c)
        mm(self.mid_block.motion_modules[0], c, sizes[-1], nlev - 1)
        for i, blk in enumerate(self.up_blocks):
            c = list(reversed(cfg.block_out_channels))[i]
            for j in range(len(blk.resnets)):
                if blk.has_cross_attention:
                    t2d(blk.attentions[j], c)
                mm(blk.motion_modules[j], c, sizes[-(i + 1)], nlev - 1 - i)

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, camera=None,
                num_views: int = 4, i2v_cond_time_zero: bool = False, return_dict: bool = True,
                down_block_additional_residuals=None, mid_block_additional_residual=None, **unused):
        assert sample.shape[0] % num_views == 0, "[UNet] input batch size must be dividable by num_views!"   # :684
        V, _, num_frames, h, w = sample.shape
        timesteps = timestep
        if not torch.is_tensor(timesteps):                                  # :706-717
            timesteps = torch.tensor([timesteps], dtype=torch.float64 if isinstance(timestep, float) else torch.int64)
        elif timesteps.ndim == 0:
            timesteps = timesteps[None]
        timesteps = timesteps.expand(V)                                     # :721
        dim0 = self.cfg.block_out_channels[0]
        emb = self.time_embedding(timestep_sinusoid(timesteps, dim0).to(sample.dtype))      # :723-730
        if i2v_cond_time_zero:                                              # :732-737
            cond_emb = self.time_embedding(timestep_sinusoid(torch.zeros(V), dim0).to(sample.dtype))
        if camera is not None:                                              # :740-745
            assert camera.shape[0] == emb.shape[0]
            cam = self.camera_embedding(camera)
            emb = emb + cam
            if i2v_cond_time_zero:
                cond_emb = cond_emb + cam
        emb = emb.repeat_interleave(num_frames, dim=0)                      # :747
        if i2v_cond_time_zero:                                              # :748-752
            emb = emb.reshape(V, num_frames, -1)
            emb = torch.cat([cond_emb[:, None], emb[:, 1:]], dim=1).reshape(V * num_frames, -1)
        ehs = encoder_hidden_states.repeat_interleave(num_frames, dim=0)    # :754
        if self.encoder_hid_proj is not None and self.cfg.encoder_hid_dim_type == "ip_image_proj":   # :756-764
            if added_cond_kwargs is None or "image_embeds" not in added_cond_kwargs:
                raise ValueError("encoder_hid_dim_type 'ip_image_proj' requires added_cond_kwargs['image_embeds']")
            ip = self.encoder_hid_proj(added_cond_kwargs["image_embeds"])
            ehs = (ehs, [t.repeat_interleave(num_frames, dim=0) for t in ip])
        else:
            ehs = (ehs, [])
        x = sample.permute(0, 2, 1, 3, 4).reshape(V * num_frames, -1, h, w)  # :767
        x = self.conv_in(x)
        skips = (x,)
        for blk in self.down_blocks:                                         # :771-785
            x, outs = blk(x, emb, ehs, num_frames)
            skips += outs
        if down_block_additional_residuals is not None:                      # :787-796 (ControlNet)
            skips = tuple(s_ + r_ for s_, r_ in zip(skips, down_block_additional_residuals))
        x = self.mid_block(x, emb, ehs, num_frames)                          # :799-815
        if mid_block_additional_residual is not None:                        # :816-817
            x = x + mid_block_additional_residual
        forward_upsample_size = any(s % (2 ** self.num_upsamplers) != 0 for s in (h, w))   # :690-698
        for i, blk in enumerate(self.up_blocks):                             # :823-852
            k = len(blk.resnets)
            res, skips = skips[-k:], skips[:-k]
            up_size = skips[-1].shape[2:] if (forward_upsample_size and i != len(self.up_blocks) - 1) else None
            x = blk(x, res, emb, ehs, num_frames, up_size)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))                     # :855-859
        x = x.reshape(V, num_frames, -1, h, w).permute(0, 2, 1, 3, 4)        # :862
        if not return_dict:
            return (x,)
        return SimpleNamespace(sample=x)


# --------------------------------------------------------------------------------------
# synthetic weights / inputs
# --------------------------------------------------------------------------------------
def init_synthetic_weights(model: nn.Module, seed: int = 0, dense: bool = True):
    """Seeded init.  Default torch init everywhere (Kaiming-uniform Linear/Conv, GN/LN gamma=1
    beta=0); with ``dense`` the zero-initialised branches get N(0, 0.02) weights so that every
    branch of the path contributes to the output (SURVEY.md §8d "dense variant")."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim >= 2:
                fan_in = p[0].numel()
                bound = 1.0 / math.sqrt(fan_in)
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
            elif name.endswith("mix_factor"):
                p.zero_()
            elif "norm" in name.split(".")[-2] and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * (torch.rand(p.shape, generator=g) * 2 - 1))
            elif name.endswith("bias"):
                p.copy_(0.02 * (torch.rand(p.shape, generator=g) * 2 - 1))
            else:
                p.copy_(1.0 + 0.1 * (torch.rand(p.shape, generator=g) * 2 - 1))
        if not dense:
            for name, p in model.named_parameters():
                if "to_out_i2v" in name:
                    p.zero_()
        else:
            for name, p in model.named_parameters():
                if "to_out_i2v.weight" in name or "to_out_sp.weight" in name:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
                if name.endswith("mix_factor"):
                    p.copy_(torch.rand(p.shape, generator=g) - 0.5)
    return model


def get_camera(num_views: int, elevation: float = 15.0, azimuth_start: float = 0.0, azimuth_span: float = 360.0):
    """Restates pipeline.py:127-190 (get_camera / generate_c2w / normalize_camera) -> [n, 16]."""
    cams = []
    for i in range(num_views):
        az = math.radians(azimuth_start + i * azimuth_span / num_views)
        el = math.radians(elevation)
        pos = torch.tensor([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)], dtype=torch.float32)
        up = torch.tensor([0.0, 0.0, 1.0])
        lookat = F.normalize(-pos, dim=0)
        right = F.normalize(torch.linalg.cross(lookat, up), dim=0)
        up = F.normalize(torch.linalg.cross(right, lookat), dim=0)
        c2w = torch.zeros(4, 4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, -lookat, pos
        c2w[3, 3] = 1.0
        t = c2w[:3, 3]
        c2w[:3, 3] = t / (torch.norm(t) + 1e-8)
        cams.append(c2w.flatten())
    return torch.stack(cams, 0).float()


def synthetic_inputs(cfg: UNetConfig, videos: int, num_views: int, num_frames: int, latent_hw, seed: int = 1,
                     cfg_doubled: bool = False):
    """Seeded synthetic call inputs in the pipeline's layout (pipeline.py:1008-1020)."""
    g = torch.Generator().manual_seed(seed)
    h, w = latent_hw
    sample = torch.randn(videos, cfg.in_channels, num_frames, h, w, generator=g)
    sample[:, :, 0] *= 0.18215
    ehs = torch.randn(videos, 77, cfg.cross_attention_dim, generator=g)
    img = torch.randn(videos, cfg.ip_image_embed_dim, generator=g)
    if cfg_doubled:
        img[: videos // 2] = 0.0          # pipeline order (uncond, text): pipeline.py:937
    cam = get_camera(num_views).repeat(videos // num_views, 1)
    return dict(sample=sample, timestep=501, encoder_hidden_states=ehs,
                added_cond_kwargs={"image_embeds": img}, camera=cam, num_views=num_views)


def build_dense(cfg: UNetConfig, num_views: int, num_frames: int, latent_hw, seed: int = 0, dense: bool = True,
                state_dict=None) -> "MVUNetMotionModelRef":
    """``MVUNetMotionModelRef(...)`` + ``init_synthetic_weights(seed, dense)`` without paying torch's default parameter init
    (~20 s of single-threaded Kaiming draws at the SD1.5 widths, all of it overwritten): build on the meta device, materialise,
    refill the positional buffers, then draw the same seeded weights — value for value what the two-step construction gives.
    With ``state_dict`` the tensors of an already drawn model are adopted instead (shared storage, read-only use): another
    token geometry over the same weights costs ~1 s."""
    with torch.device("meta"):
        m = MVUNetMotionModelRef(cfg, num_views, num_frames, latent_hw)
    if state_dict is not None:
        m.load_state_dict(state_dict, strict=True, assign=True)
        return m.eval()
    m = m.to_empty(device="cpu").eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, TimePosEmbed):
                mod.pe.copy_(sinusoidal_pos_1d(mod.pe.shape[2], mod.pe.shape[1]))
    return init_synthetic_weights(m, seed=seed, dense=dense)


def build_fast(cfg: UNetConfig, num_views: int, num_frames: int, latent_hw, seed: Optional[int] = 0) -> "MVUNetMotionModelRef":
    """Construct the oracle without torch's default (slow, single-threaded) parameter init: build on the
    meta device, materialise, refill the positional buffers and draw cheap seeded uniform weights.
    Used by bench.py's cpu_baseline leg and smoke(), where only timing / same-weights parity matter."""
    with torch.device("meta"):
        m = MVUNetMotionModelRef(cfg, num_views, num_frames, latent_hw)
    m = m.to_empty(device="cpu").eval()
    if seed is None:      # timing only: constant fill (values do not change the FLOPs; avoids drawing 1.5e9 randoms)
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, TimePosEmbed):
                    mod.pe.copy_(sinusoidal_pos_1d(mod.pe.shape[2], mod.pe.shape[1]))
            for name, p in m.named_parameters():
                p.fill_(1.0 / math.sqrt(p[0].numel()) / 3 if p.ndim >= 2 else (0.0 if name.endswith(("bias", "mix_factor")) else 1.0))
        return m
    torch.manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, TimePosEmbed):
                mod.pe.copy_(sinusoidal_pos_1d(mod.pe.shape[2], mod.pe.shape[1]))
        for name, p in m.named_parameters():
            if p.ndim >= 2:
                b = 1.0 / math.sqrt(p[0].numel())
                p.uniform_(-b, b)
            elif name.endswith("mix_factor"):
                p.uniform_(-0.5, 0.5)
            elif name.endswith("bias"):
                p.uniform_(-0.02, 0.02)
            else:
                p.uniform_(0.9, 1.1)
        for name, p in m.named_parameters():
            if name.endswith("to_out_i2v.weight") or name.endswith("to_out_sp.weight"):
                p.normal_(0.0, 0.02)
    return m
