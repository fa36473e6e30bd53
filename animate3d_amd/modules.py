"""Parameter tree of the MV-VDM UNet under the reference's (diffusers 0.28) state-dict names.

These classes only HOLD parameters — none of them has a forward.  The arithmetic is issued by
``animate3d_amd.unet`` through the HIP op set; the tree exists so that
``load_state_dict(reference_checkpoint)`` / ``state_dict()`` / ``attn_processors`` /
``set_attn_processor`` behave like the reference model's
(animatediff/models/unet_motion_mv_model.py:439-497, inference.py:90-223; key families listed
in SURVEY.md Appendix A.8).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .config import UNetConfig
from .embeddings import sinusoidal_pos_1d


class Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the forward pass is issued by animate3d_amd.unet.MVUNetMotionModel")


def _lin(i, o, bias=True):
    return nn.Linear(i, o, bias=bias)


# ---------------- attention processors (reference: animatediff/models/attention_processor.py)
class MVDreamAttnProcessor(Holder):
    """Multi-view self-attention without the I2V branch (reference class
    MVDreamXFormersAttnProcessor, attention_processor.py:22-126).  No parameters."""
    kind = "mvdream"


class MVDreamI2VAttnProcessor(Holder):
    """Multi-view self-attention + first-frame (I2V) attention (reference class
    MVDreamI2VXFormersAttnProcessor, attention_processor.py:302-445)."""
    kind = "mvdream_i2v"

    def __init__(self, hidden_size: int):
        super().__init__()
        self.hidden_size = hidden_size
        self.to_q_i2v = _lin(hidden_size, hidden_size, bias=False)
        self.to_out_i2v = _lin(hidden_size, hidden_size, bias=True)


class IPAdapterAttnProcessor(Holder):
    """Text + IP-Adapter image cross-attention (reference class IPAdapterXFormersAttnProcessor,
    attention_processor.py:129-298)."""
    kind = "ip_adapter"

    def __init__(self, hidden_size: int, cross_attention_dim: int, num_tokens=(4,), scale: float = 1.0):
        super().__init__()
        self.hidden_size, self.cross_attention_dim = hidden_size, cross_attention_dim
        self.num_tokens = list(num_tokens)
        self.scale = [scale] * len(self.num_tokens)
        self.to_k_ip = nn.ModuleList([_lin(cross_attention_dim, hidden_size, bias=False) for _ in self.num_tokens])
        self.to_v_ip = nn.ModuleList([_lin(cross_attention_dim, hidden_size, bias=False) for _ in self.num_tokens])


class _PE(Holder):
    def __init__(self, dim, max_len):
        super().__init__()
        self.register_buffer("pe", sinusoidal_pos_1d(dim, max_len))


class _AlphaBlender(Holder):
    def __init__(self):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.zeros(1))


class _Learned2D(Holder):
    """embeddings.py:99-157 LearnedPositionalEncoding2D: channels [0, C/2) = col_embed(x), [C/2, C) = row_embed(y)."""

    def __init__(self, num_feats, rows, cols):
        super().__init__()
        self.row_embed = nn.Embedding(rows, num_feats)
        self.col_embed = nn.Embedding(cols, num_feats)
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)


class _LabelEmbedding(Holder):
    def __init__(self, num_classes, hidden_size):
        super().__init__()
        self.embedding_table = nn.Embedding(num_classes, hidden_size)


class _SoftmaxAlphaBlender(Holder):
    def __init__(self):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.zeros(3))


class SpatioTemporalI2VAttnProcessor(Holder):
    """Motion-module processor: temporal attention + optional multi-view spatial attention (2-D positional encoding
    sinusoid / learnable, optional per-view camera encoding) + optional first-frame image attention, merged by sum,
    AlphaBlender or the 3-way SoftmaxAlphaBlender (reference class SpatioTemporalI2VXFormersAttnProcessor,
    attention_processor.py:448-744).  Released switch set: spatial attention with the sinusoid encoding and the blender."""
    kind = "spatio_temporal"

    def __init__(self, hidden_size: int, spatial_attn: bool = True, use_spatial_encoding: bool = True,
                 use_alpha_blender: bool = True, max_seq_length: int = 32, image_attn: bool = False,
                 use_camera_encoding: bool = False, spatial_encoding_type: str = "sinusoid",
                 camera_encoding_type: str = "sinusoid", feature_size: int = 32, num_views: Optional[int] = None):
        super().__init__()
        self.hidden_size = hidden_size
        self.use_spatial_attn, self.use_spatial_encoding, self.use_alpha_blender = spatial_attn, use_spatial_encoding, use_alpha_blender
        self.use_image_attn, self.use_camera_encoding = image_attn, use_camera_encoding
        self.spatial_encoding_type, self.camera_encoding_type = spatial_encoding_type, camera_encoding_type
        if spatial_attn:
            self.to_q_sp = _lin(hidden_size, hidden_size, bias=False)
            self.to_k_sp = _lin(hidden_size, hidden_size, bias=False)
            self.to_v_sp = _lin(hidden_size, hidden_size, bias=False)
            self.to_out_sp = _lin(hidden_size, hidden_size, bias=True)
            if use_spatial_encoding or use_camera_encoding:
                self.time_pos_embed = _PE(hidden_size, max_seq_length)
            if use_spatial_encoding:
                if spatial_encoding_type == "learnable":
                    self.spatial_pos_embed = _Learned2D(hidden_size // 2, feature_size, feature_size)
                elif spatial_encoding_type != "sinusoid":
                    raise ValueError(f"Spatial encoding type {spatial_encoding_type} is not supported yet!")
            if use_camera_encoding:
                if num_views is None:
                    raise ValueError("camera encoding needs num_views at construction (MVUNetMotionModel(num_views=...))")
                if camera_encoding_type == "learnable":
                    self.camera_embed = _LabelEmbedding(num_views, hidden_size)
                elif camera_encoding_type == "sinusoid":
                    self.camera_embed = _PE(hidden_size, num_views)
                else:
                    raise ValueError(f"Camera encoding type {camera_encoding_type} is not supported yet!")
        if image_attn:
            self.to_q_i2v = _lin(hidden_size, hidden_size, bias=False)
            self.to_k_i2v = _lin(hidden_size, hidden_size, bias=False)
            self.to_v_i2v = _lin(hidden_size, hidden_size, bias=False)
            self.to_out_i2v = _lin(hidden_size, hidden_size, bias=True)
        num_attn = 1 + int(spatial_attn) + int(image_attn)
        if not use_alpha_blender:
            for m in ([self.to_out_sp] if spatial_attn else []) + ([self.to_out_i2v] if image_attn else []):
                nn.init.zeros_(m.weight)
                nn.init.zeros_(m.bias)
        elif num_attn == 2:
            self.alpha_blender = _AlphaBlender()
        elif num_attn == 3:
            self.alpha_blender = _SoftmaxAlphaBlender()

    def blend_coefficients(self):
        """(temporal, spatial, image) weights of the merge (attention_processor.py:700-713, 727-744)."""
        sp, im = self.use_spatial_attn, self.use_image_attn
        if not self.use_alpha_blender or not (sp or im):
            return 1.0, 1.0 if sp else 0.0, 1.0 if im else 0.0
        m = self.alpha_blender.mix_factor.detach().float()
        if sp and im:
            a = torch.softmax(m, dim=0)
            return float(a[1]), float(a[0]), float(a[2])
        a = float(torch.sigmoid(m)[0])
        return 1.0 - a, (a if sp else 0.0), (a if im else 0.0)

    def blend_coefficients_t(self):
        """Same weights with autograd history (training path): 0-dim tensors where ``mix_factor`` is involved."""
        sp, im = self.use_spatial_attn, self.use_image_attn
        if not self.use_alpha_blender or not (sp or im):
            return self.blend_coefficients()
        m = self.alpha_blender.mix_factor.float()
        if sp and im:
            a = torch.softmax(m, dim=0)
            return a[1], a[0], a[2]
        a = torch.sigmoid(m)[0]
        return 1.0 - a, (a if sp else 0.0), (a if im else 0.0)


# ---------------- diffusers-named containers
class Attention(Holder):
    def __init__(self, query_dim: int, cross_attention_dim: Optional[int], heads: int, dim_head: int):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = _lin(query_dim, inner, bias=False)
        self.to_k = _lin(kv, inner, bias=False)
        self.to_v = _lin(kv, inner, bias=False)
        self.to_out = nn.ModuleList([_lin(inner, query_dim, bias=True), nn.Dropout(0.0)])
        self.processor = None

    def set_processor(self, processor):
        self.processor = processor

    def get_processor(self, return_deprecated_lora: bool = False):
        return self.processor


class GEGLU(Holder):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = _lin(dim_in, dim_out * 2)


class FeedForward(Holder):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), _lin(dim * mult, dim)])


class BasicTransformerBlock(Holder):
    def __init__(self, dim, heads, head_dim, cross_attention_dim, double_self_attention=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, heads, head_dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, None if double_self_attention else cross_attention_dim, heads, head_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)
        self.pos_embed = None       # inference.py:176-192 nulls it; the processor owns the temporal PE


class Transformer2DModel(Holder):
    def __init__(self, heads, head_dim, in_channels, cross_attention_dim, groups):
        super().__init__()
        inner = heads * head_dim
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, head_dim, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)


class TransformerTemporalModel(Holder):
    def __init__(self, heads, head_dim, in_channels, groups):
        super().__init__()
        inner = heads * head_dim
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = _lin(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, head_dim, None, double_self_attention=True)])
        self.proj_out = _lin(inner, in_channels)


class ResnetBlock2D(Holder):
    def __init__(self, in_c, out_c, temb_c, groups, eps):
        super().__init__()
        self.in_channels, self.out_channels = in_c, out_c
        self.norm1 = nn.GroupNorm(groups, in_c, eps=eps)
        self.conv1 = nn.Conv2d(in_c, out_c, 3, padding=1)
        self.time_emb_proj = _lin(temb_c, out_c)
        self.norm2 = nn.GroupNorm(groups, out_c, eps=eps)
        self.conv2 = nn.Conv2d(out_c, out_c, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_c, out_c, 1) if in_c != out_c else None


class Sampler2D(Holder):
    """Downsample2D (3x3 stride-2 conv) / Upsample2D (nearest 2x + 3x3 conv): both hold ``conv``."""

    def __init__(self, channels, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=stride, padding=1)


class TimestepEmbedding(Holder):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = _lin(in_dim, dim)
        self.linear_2 = _lin(dim, dim)


class ImageProjection(Holder):
    def __init__(self, image_embed_dim, cross_attention_dim, num_tokens):
        super().__init__()
        self.num_image_text_embeds = num_tokens
        self.image_embeds = _lin(image_embed_dim, num_tokens * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)


class MultiIPAdapterImageProjection(Holder):
    def __init__(self, layers):
        super().__init__()
        self.image_projection_layers = nn.ModuleList(layers)


class MotionBlock(Holder):
    """{CrossAttn}DownBlockMotion / UNetMidBlockCrossAttnMotion / {CrossAttn}UpBlockMotion."""

    def __init__(self, cfg: UNetConfig, kind: str, resnet_io, out_c: int, temb_c: int, has_attn: bool, n_attn: int,
                 n_motion: int, sampler: Optional[str]):
        super().__init__()
        self.kind, self.has_cross_attention = kind, has_attn
        g, eps = cfg.norm_num_groups, cfg.norm_eps
        self.resnets = nn.ModuleList([ResnetBlock2D(i, o, temb_c, g, eps) for i, o in resnet_io])
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cfg.num_attention_heads, out_c // cfg.num_attention_heads, out_c,
                                                                cfg.cross_attention_dim, g) for _ in range(n_attn)])
        self.motion_modules = nn.ModuleList([TransformerTemporalModel(cfg.motion_num_attention_heads, out_c // cfg.motion_num_attention_heads,
                                                                      out_c, g) for _ in range(n_motion)])
        self.downsamplers = nn.ModuleList([Sampler2D(out_c, 2)]) if sampler == "down" else None
        self.upsamplers = nn.ModuleList([Sampler2D(out_c, 1)]) if sampler == "up" else None
