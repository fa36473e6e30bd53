"""Constructor constants of the MV-VDM UNet (reference: MVUNetMotionModel.__init__,
animatediff/models/unet_motion_mv_model.py:67-102) plus the processor switches of
configs/inference/inference.yaml:9-24, with the released values as defaults."""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Optional, Tuple


@dataclass
class UNetConfig:
    sample_size: Optional[int] = 32
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)   # CrossAttnDownBlockMotion x3, DownBlockMotion
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    cross_attention_dim: int = 768
    num_attention_heads: int = 8
    motion_num_attention_heads: int = 8
    motion_max_seq_length: int = 32
    camera_embedding_dim: Optional[int] = 16
    ip_image_embed_dim: Optional[int] = 1024
    ip_num_tokens: int = 4
    ip_scale: float = 1.0
    mvdream_image_attn: bool = True           # mvdream_attn_cfg.image_attn.enabled
    motion_spatial_attn: bool = True          # motion_module_attn_cfg.spatial_attn.enabled
    motion_use_spatial_encoding: bool = True  # ...attn_cfg.use_spatial_encoding (sinusoid)
    motion_use_alpha_blender: bool = True     # motion_module_attn_cfg.use_alpha_blender
    # switches the released configs leave off (attention_processor.py:478-540)
    motion_image_attn: bool = False           # motion_module_attn_cfg.image_attn.enabled: first-frame attention per view
    motion_use_camera_encoding: bool = False  # ...spatial_attn.attn_cfg.use_camera_encoding: one vector per view
    motion_spatial_encoding_type: str = "sinusoid"    # or "learnable" (embeddings.py:99-157)
    motion_camera_encoding_type: str = "sinusoid"     # or "learnable" (LabelEmbedding table)
    encoder_hid_dim_type: Optional[str] = "ip_image_proj"

    def to_dict(self):
        return asdict(self)

    # diffusers-style attribute/dict access used by callers (pipeline.py:880-881,955)
    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, default=None):
        return getattr(self, k, default)
