"""The condition towers in front of the denoising loop — SURVEY.md §8(f) item 3, first half.

``animatediff/pipelines/pipeline.py:345-524`` (``encode_prompt``: ``text_encoder(ids)[0]``) and ``:527-538`` /
``animatediff/utils/util.py:268-287`` (``encode_image``: ``image_encoder(pixels).image_embeds`` and a zero tensor for the
unconditional half) call two third-party transformers models once per sample:

* the SD1.5 text encoder, ``transformers.CLIPTextModel`` (ViT-L/14 text tower: 12 layers x 768, 12 heads of 64, QuickGELU,
  causal attention over 77 tokens, final LayerNorm) -> ``prompt_embeds [B, 77, 768]``;
* the IP-Adapter image encoder, ``transformers.CLIPVisionModelWithProjection`` (ViT-H/14: 32 layers x 1280, 16 heads of 80,
  GELU, 257 tokens, post-LayerNorm of the class token, 1280 -> 1024 projection) -> ``image_embeds [B, 1024]``.

Both run here on the kernels of the denoise step (LayerNorm, fused-QKV GEMMs with bias, flash attention — head dim 64 with the
causal flag, head dim 80 —, activation, residual GEMM epilogues) under transformers' parameter names, so
``load_state_dict(hf_model.state_dict())`` works key for key.  Tokenisation and image pre-processing (CLIPTokenizer,
CLIPImageProcessor) stay with the caller: inputs are token ids and normalised pixel values.  No CPU fallback.
Parity: the oracle is transformers' own implementation (installed here; architecture unchanged since the reference's pin
4.25.1), tests/test_clip.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .hip_ops import RowMap, on_model_device
from .modules import Holder


@dataclass
class CLIPTowerConfig:
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    # text
    vocab_size: int = 49408
    max_position_embeddings: int = 77
    # vision
    image_size: int = 224
    patch_size: int = 14
    num_channels: int = 3
    projection_dim: int = 1024


TEXT_SD15 = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, hidden_act="quick_gelu")
VISION_VIT_H = dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, hidden_act="gelu",
                    image_size=224, patch_size=14, projection_dim=1024)


class _Attn(Holder):
    def __init__(self, c):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)


class _MLP(Holder):
    def __init__(self, c, i):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(c, i), nn.Linear(i, c)


class _Layer(Holder):
    def __init__(self, cfg):
        super().__init__()
        c = cfg.hidden_size
        self.self_attn = _Attn(c)
        self.layer_norm1 = nn.LayerNorm(c, eps=cfg.layer_norm_eps)
        self.mlp = _MLP(c, cfg.intermediate_size)
        self.layer_norm2 = nn.LayerNorm(c, eps=cfg.layer_norm_eps)


class _Encoder(Holder):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])


class _TextEmbeddings(Holder):
    def __init__(self, cfg):
        super().__init__()
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.position_embedding = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)


class _TextTransformer(Holder):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _TextEmbeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.final_layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class _VisionEmbeddings(Holder):
    def __init__(self, cfg):
        super().__init__()
        c = cfg.hidden_size
        self.class_embedding = nn.Parameter(torch.randn(c))
        self.patch_embedding = nn.Conv2d(cfg.num_channels, c, cfg.patch_size, stride=cfg.patch_size, bias=False)
        self.position_embedding = nn.Embedding((cfg.image_size // cfg.patch_size) ** 2 + 1, c)


class _VisionTransformer(Holder):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _VisionEmbeddings(cfg)
        self.pre_layrnorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)          # (sic: transformers' spelling)
        self.encoder = _Encoder(cfg)
        self.post_layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class _Tower(nn.Module):
    """Shared machinery: op set following the model dtype, packed weights, one pre-LN transformer layer."""

    def __init__(self, config: CLIPTowerConfig, ops=None):
        super().__init__()
        self.config = config
        self._ops, self._ops_auto, self._packed = ops, False, None

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

    def _apply(self, fn, *a, **k):
        self._packed = None
        if self._ops_auto:
            self._ops, self._ops_auto = None, False
        return super()._apply(fn, *a, **k)

    _PREFIX = ""          # "text_model." / "vision_model.": transformers <= 4.x key prefix (the reference's checkpoints have it)

    def load_state_dict(self, sd, strict: bool = True, assign: bool = False):
        sd = {k: v for k, v in sd.items() if not k.endswith("position_ids")}        # buffer of older transformers versions
        fam = ("embeddings.", "encoder.", "final_layer_norm.", "pre_layrnorm.", "post_layernorm.")
        sd = {(self._PREFIX + k if k.startswith(fam) else k): v for k, v in sd.items()}   # transformers 5.x dropped the prefix
        self._packed = None
        return super().load_state_dict(sd, strict=strict, assign=assign)

    def _w(self, t):
        return t.detach().to(self.ops.act_dtype).contiguous()

    @staticmethod
    def _f(t):
        return t.detach().float().contiguous()

    def _pack_layers(self, enc: _Encoder):
        out = []
        for l in enc.layers:
            a = l.self_attn
            out.append(SimpleNamespace(
                n1=(self._f(l.layer_norm1.weight), self._f(l.layer_norm1.bias)),
                qkv=(self._w(torch.cat([a.q_proj.weight.detach(), a.k_proj.weight.detach(), a.v_proj.weight.detach()], 0)),
                     self._f(torch.cat([a.q_proj.bias.detach(), a.k_proj.bias.detach(), a.v_proj.bias.detach()], 0))),
                o=(self._w(a.out_proj.weight), self._f(a.out_proj.bias)),
                n2=(self._f(l.layer_norm2.weight), self._f(l.layer_norm2.bias)),
                fc1=(self._w(l.mlp.fc1.weight), self._f(l.mlp.fc1.bias)), fc2=(self._w(l.mlp.fc2.weight), self._f(l.mlp.fc2.bias))))
        return out

    def _layer(self, x, pk, B, T, causal):
        """transformers CLIPEncoderLayer: x += attn(LN1(x)); x += fc2(act(fc1(LN2(x))))."""
        ops, cfg = self.ops, self.config
        C = x.shape[1]
        h = ops.layer_norm(x, pk.n1[0], pk.n1[1], cfg.layer_norm_eps)
        qkv = ops.gemm(h, pk.qkv[0], pk.qkv[1])
        m = RowMap(gdiv=1, ga=T, gb=0, seg_len=T, seg_stride=0)
        a = ops.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], m, m, B, cfg.num_attention_heads, T, T, causal=causal)
        x = ops.gemm(a, pk.o[0], pk.o[1], residual=x)
        h = ops.layer_norm(x, pk.n2[0], pk.n2[1], cfg.layer_norm_eps)
        h = ops.activation(ops.gemm(h, pk.fc1[0], pk.fc1[1]), cfg.hidden_act)
        return ops.gemm(h, pk.fc2[0], pk.fc2[1], residual=x)


class CLIPTextEncoder(_Tower):
    """``transformers.CLIPTextModel`` of the SD1.5 pipeline (inference.py:64) — ``forward(input_ids)[0]`` = last hidden state."""
    _PREFIX = "text_model."

    def __init__(self, config: Optional[CLIPTowerConfig] = None, ops=None, device: Optional[Union[str, torch.device]] = None, **overrides):
        cfg = config if config is not None else CLIPTowerConfig(**{**TEXT_SD15, **overrides})
        super().__init__(cfg, ops)
        with (torch.device(device) if device is not None else torch.device("cpu")):
            self.text_model = _TextTransformer(cfg)

    def _pack(self):
        tm = self.text_model
        self._packed = SimpleNamespace(layers=self._pack_layers(tm.encoder),
                                       final=(self._f(tm.final_layer_norm.weight), self._f(tm.final_layer_norm.bias)))
        return self._packed

    @torch.no_grad()
    @on_model_device
    def forward(self, input_ids: torch.Tensor, attention_mask=None, **unused) -> Tuple[torch.Tensor]:
        if attention_mask is not None:
            raise NotImplementedError("the reference passes attention_mask=None (SD1.5 text encoder config has no use_attention_mask)")
        P = self._packed if self._packed is not None else self._pack()
        ops, cfg, emb = self.ops, self.config, self.text_model.embeddings
        B, T = input_ids.shape
        if T > cfg.max_position_embeddings:
            raise ValueError(f"{T} tokens exceed max_position_embeddings = {cfg.max_position_embeddings}")
        ids = input_ids.to(self.device)
        x = (emb.token_embedding.weight[ids].float() + emb.position_embedding.weight[:T].float()[None]).reshape(B * T, -1)
        x = x.to(ops.act_dtype).contiguous()
        for pk in P.layers:
            x = self._layer(x, pk, B, T, causal=True)
        x = ops.layer_norm(x, P.final[0], P.final[1], cfg.layer_norm_eps)
        return (x.reshape(B, T, -1).to(self.dtype if self.dtype != torch.float32 else torch.float32),)


class CLIPVisionEncoderWithProjection(_Tower):
    """``transformers.CLIPVisionModelWithProjection`` (IP-Adapter image encoder, inference.py:78): ``forward(pixels).image_embeds``."""
    _PREFIX = "vision_model."

    def __init__(self, config: Optional[CLIPTowerConfig] = None, ops=None, device: Optional[Union[str, torch.device]] = None, **overrides):
        cfg = config if config is not None else CLIPTowerConfig(**{**VISION_VIT_H, **overrides})
        super().__init__(cfg, ops)
        with (torch.device(device) if device is not None else torch.device("cpu")):
            self.vision_model = _VisionTransformer(cfg)
            self.visual_projection = nn.Linear(cfg.hidden_size, cfg.projection_dim, bias=False)

    def _pack(self):
        vm, cfg = self.vision_model, self.config
        k = cfg.num_channels * cfg.patch_size ** 2
        kp = (k + 63) // 64 * 64                              # the GEMM contracts in steps of 64: zero-padded patch vectors
        wp = torch.zeros(cfg.hidden_size, kp, device=self.device, dtype=torch.float32)
        wp[:, :k] = vm.embeddings.patch_embedding.weight.detach().float().reshape(cfg.hidden_size, k)       # (c, ky, kx) order
        self._packed = SimpleNamespace(
            patch=self._w(wp), kp=kp, pre=(self._f(vm.pre_layrnorm.weight), self._f(vm.pre_layrnorm.bias)),
            layers=self._pack_layers(vm.encoder), post=(self._f(vm.post_layernorm.weight), self._f(vm.post_layernorm.bias)),
            proj=self._w(self.visual_projection.weight))
        return self._packed

    @torch.no_grad()
    @on_model_device
    def forward(self, pixel_values: torch.Tensor, **unused):
        P = self._packed if self._packed is not None else self._pack()
        ops, cfg, emb = self.ops, self.config, self.vision_model.embeddings
        B, Cc, H, W = pixel_values.shape
        ps = cfg.patch_size
        gh, gw = H // ps, W // ps
        if (gh * gw + 1) != emb.position_embedding.weight.shape[0]:
            raise ValueError(f"image {H}x{W} does not match the position table ({emb.position_embedding.weight.shape[0]} entries)")
        # non-overlapping patches are a pure re-layout: [B, C, gh, ps, gw, ps] -> [(B gh gw), (C ps ps)], zero-padded to the GEMM step
        px = pixel_values.to(self.device).float().reshape(B, Cc, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, Cc * ps * ps)
        pad = torch.zeros(B * gh * gw, P.kp, device=self.device, dtype=ops.act_dtype)
        pad[:, : px.shape[1]] = px.to(ops.act_dtype)
        patches = ops.gemm(pad, P.patch).reshape(B, gh * gw, -1)
        T = gh * gw + 1
        x = torch.cat([emb.class_embedding.to(patches.dtype)[None, None].expand(B, 1, -1), patches], dim=1)
        x = (x.float() + emb.position_embedding.weight[:T].float()[None]).to(ops.act_dtype).reshape(B * T, -1).contiguous()
        x = ops.layer_norm(x, P.pre[0], P.pre[1], cfg.layer_norm_eps)
        for pk in P.layers:
            x = self._layer(x, pk, B, T, causal=False)
        pooled = ops.layer_norm(x.reshape(B, T, -1)[:, 0].contiguous(), P.post[0], P.post[1], cfg.layer_norm_eps)
        embeds = ops.gemm(pooled, P.proj)
        out_dtype = self.dtype
        return SimpleNamespace(image_embeds=embeds.to(out_dtype), last_hidden_state=x.reshape(B, T, -1).to(out_dtype))


@torch.no_grad()
def encode_prompt(text_encoder, input_ids: torch.Tensor, negative_input_ids: Optional[torch.Tensor] = None):
    """pipeline.py:345-524 without the tokenizer: (prompt_embeds, negative_prompt_embeds or None), each [B, 77, 768]."""
    pe = text_encoder(input_ids)[0]
    ne = text_encoder(negative_input_ids)[0] if negative_input_ids is not None else None
    return pe, ne


@torch.no_grad()
def encode_image(image_encoder, pixel_values: torch.Tensor):
    """pipeline.py:527-538: (image_embeds, zeros_like(image_embeds)) — the unconditional half is all zeros."""
    e = image_encoder(pixel_values).image_embeds
    return e, torch.zeros_like(e)
