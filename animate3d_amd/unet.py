"""MI355X-native MV-VDM denoising UNet behind the reference's call surface.

``MVUNetMotionModel`` here is a drop-in for the reference class of the same name
(animatediff/models/unet_motion_mv_model.py:55-867): same ``forward`` signature and return type,
same parameter names (a reference / diffusers state-dict loads key-for-key), same
``attn_processors`` / ``set_attn_processor`` / ``config`` / ``dtype`` / ``device`` surface that
``animatediff/pipelines/pipeline.py:1012-1020`` and the 4D-SDS guidance
(``custom/threestudio-animate3d/guidance/animatemv_guidance.py:339-346``) touch.

What is different is everything underneath:

* activations are token-major NHWC rows ``[(b n f) h w, C]`` in bf16 for the whole step, so the
  reference's three token groupings — per image, per multi-view group ``(b f)(n l)``, per pixel
  sequence ``(b n h w) f`` (attention_processor.py:54,340,552-557) — are addressing modes of the
  attention kernels, not ``rearrange(...).contiguous()`` copies;
* every arithmetic op is a hand-written gfx950 kernel reached through the C-ABI of
  ``libanimate3d_hip.so`` (include/animate3d_hip.h) — there is no eager/PyTorch fallback;
* Q/K/V(/Q_i2v) projections are single fused GEMMs; residual adds, the AlphaBlender mix, the
  time-embedding broadcast and the I2V sum are GEMM/conv epilogues; text / IP-Adapter K/V are
  projected once per video instead of once per frame (the reference repeats them F times,
  unet_motion_mv_model.py:754,763);
* token geometry (views, frames, feature size per level) is derived from the call, not from
  processor constructor constants (fixes SURVEY.md F5: hard-coded sample_size=256).
"""
from __future__ import annotations

import math
import os
import re
from types import SimpleNamespace
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import modules as M
from .config import UNetConfig
from .embeddings import sine_pos_2d, sinusoidal_pos_1d
from .hip_ops import RowMap, on_model_device


class UNet3DConditionOutput:
    """Same shape as diffusers' UNet3DConditionOutput: ``.sample`` is ``[V, C, F, H, W]``."""

    def __init__(self, sample: torch.Tensor):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


def _rows(w: torch.Tensor, r0: int, r1: Optional[int] = None) -> torch.Tensor:
    """Row range of a fused projection weight.  A packed TRAINABLE operand (autograd_ops.pack_weight) keeps its fp32 gradient sink."""
    r1 = w.shape[0] if r1 is None else r1
    if getattr(w, "_a3d_sink", None) is None:
        return w[r0:r1]
    from .autograd_ops import weight_rows
    return weight_rows(w, r0, r1)


class MVUNetMotionModel(nn.Module):
    def __init__(self, config: Optional[UNetConfig] = None, ops=None, num_views: Optional[int] = None,
                 device: Optional[Union[str, torch.device]] = None, **config_overrides):
        super().__init__()
        cfg = config if config is not None else UNetConfig(**config_overrides)
        self.config = cfg
        self.sample_size = cfg.sample_size
        self.num_views = num_views          # processors' view count; None => taken from forward(num_views=)
        self._ops = ops
        self._ops_auto = False
        self._packed = None
        self._packed_frozen = None          # training: persistent pack of the frozen sub-modules (enable_training)
        self._pack_grad = False             # True while _pack_train re-packs trainable sub-modules under autograd
        self._train_ops = None              # AutogradOps over the op set once enable_training() was called
        self._active_ops = None
        self._cond = None                   # inference: the stacked conditioning projections of the running forward (_project_conditioning)
        self.stack_conditioning = True      # False: one projection per layer, as in rounds 1-5 (same-box A/B of the stacking)
        self._pe_cache: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self.parallel = None                # set by animate3d_amd.parallel.shard_unet
        self._frames = (0, 0)               # (frames of the call, first frame of this rank): set by forward
        if len(cfg.block_out_channels) != len(cfg.down_has_attn):
            raise ValueError("block_out_channels and down_has_attn must have the same length")
        if 9 * cfg.in_channels > 64:        # conv_in runs as a K = 64 GEMM over 3x3 patches (a3d_im2col_in)
            raise ValueError(f"in_channels = {cfg.in_channels} is not supported (<= 7; the MV-VDM UNet has 4 — the 9-channel PIA conv_in "
                             "of unet_motion_mv_model.py:312-330 is not on this path)")
        ctx = torch.device(device) if device is not None else torch.device("cpu")
        with ctx:
            self._build(cfg)
            self._install_default_processors()

    # ------------------------------------------------------------------ construction
    def _build(self, cfg: UNetConfig):
        boc = cfg.block_out_channels
        temb_c = boc[0] * 4
        nlev = len(boc)
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = M.TimestepEmbedding(boc[0], temb_c)
        if cfg.camera_embedding_dim is not None:
            self.camera_embedding = M.TimestepEmbedding(cfg.camera_embedding_dim, temb_c)
        self.encoder_hid_proj = None
        if cfg.encoder_hid_dim_type == "ip_image_proj":
            self.encoder_hid_proj = M.MultiIPAdapterImageProjection(
                [M.ImageProjection(cfg.ip_image_embed_dim, cfg.cross_attention_dim, cfg.ip_num_tokens)])
        n = cfg.layers_per_block
        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()      # registered before mid_block: diffusers' module order (IP-Adapter key ids)
        out_c = boc[0]
        for i in range(nlev):
            in_c, out_c = out_c, boc[i]
            io = [(in_c if j == 0 else out_c, out_c) for j in range(n)]
            self.down_blocks.append(M.MotionBlock(cfg, "down", io, out_c, temb_c, cfg.down_has_attn[i], n, n,
                                                  "down" if i != nlev - 1 else None))
        c = boc[-1]
        self.mid_block = M.MotionBlock(cfg, "mid", [(c, c), (c, c)], c, temb_c, True, 1, 1, None)
        rev, rev_attn = list(reversed(boc)), list(reversed(cfg.down_has_attn))
        out_c = rev[0]
        for i in range(nlev):
            prev_c, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, nlev - 1)]
            io = []
            for j in range(n + 1):
                skip_c = in_c if j == n else out_c
                res_in = prev_c if j == 0 else out_c
                io.append((res_in + skip_c, out_c))
            self.up_blocks.append(M.MotionBlock(cfg, "up", io, out_c, temb_c, rev_attn[i], n + 1, n + 1,
                                                "up" if i != nlev - 1 else None))
        self.num_upsamplers = nlev - 1
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, boc[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def _blocks(self):
        for i, b in enumerate(self.down_blocks):
            yield f"down_blocks.{i}", b, self.config.block_out_channels[i]
        yield "mid_block", self.mid_block, self.config.block_out_channels[-1]
        for i, b in enumerate(self.up_blocks):
            yield f"up_blocks.{i}", b, list(reversed(self.config.block_out_channels))[i]

    def _block_levels(self):
        """(block, channels, resolution level) — level l works on latents of size >> l."""
        nlev = len(self.config.block_out_channels)
        for i, b in enumerate(self.down_blocks):
            yield b, self.config.block_out_channels[i], i
        yield self.mid_block, self.config.block_out_channels[-1], nlev - 1
        for i, b in enumerate(self.up_blocks):
            yield b, list(reversed(self.config.block_out_channels))[i], nlev - 1 - i

    def _install_default_processors(self):
        """inference.py:90-174 by layer ROLE (attn1 / attn2 / motion_modules), not by diffusers class
        identity.  ``to_q_i2v := to_q`` and ``to_out_i2v := 0`` as inference.py:161-165."""
        cfg = self.config
        for blk, c, lvl in self._block_levels():
            if blk.has_cross_attention:
                for t in blk.attentions:
                    tb = t.transformer_blocks[0]
                    if cfg.mvdream_image_attn:
                        p = M.MVDreamI2VAttnProcessor(c)
                        with torch.no_grad():
                            p.to_q_i2v.weight.copy_(tb.attn1.to_q.weight)
                            p.to_out_i2v.weight.zero_()
                            p.to_out_i2v.bias.zero_()
                    else:
                        p = M.MVDreamAttnProcessor()
                    tb.attn1.set_processor(p)
                    ipp = M.IPAdapterAttnProcessor(c, cfg.cross_attention_dim, (cfg.ip_num_tokens,), cfg.ip_scale)
                    tb.attn2.set_processor(ipp)
            for m in blk.motion_modules:
                tb = m.transformer_blocks[0]
                for a in (tb.attn1, tb.attn2):
                    sp = M.SpatioTemporalI2VAttnProcessor(
                        c, cfg.motion_spatial_attn, cfg.motion_use_spatial_encoding, cfg.motion_use_alpha_blender,
                        cfg.motion_max_seq_length, image_attn=cfg.motion_image_attn, use_camera_encoding=cfg.motion_use_camera_encoding,
                        spatial_encoding_type=cfg.motion_spatial_encoding_type, camera_encoding_type=cfg.motion_camera_encoding_type,
                        feature_size=max(1, (cfg.sample_size or 32) >> lvl),      # table rows of the learnable encoding (inference.py:93-105)
                        num_views=self.num_views)
                    a.set_processor(sp)

    # ------------------------------------------------------------------ diffusers-style surface
    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def attn_processors(self) -> Dict[str, nn.Module]:
        """name -> processor, keyed exactly like the reference (unet_motion_mv_model.py:439-462)."""
        out = {}
        for name, mod in self.named_modules():
            if isinstance(mod, M.Attention):
                out[f"{name}.processor"] = mod.get_processor()
        return out

    def set_attn_processor(self, processor):
        """unet_motion_mv_model.py:465-497.  Accepts a dict name -> processor (count must match) or one
        parameter-free processor for all layers.  Processors must be animate3d_amd.modules classes."""
        names = list(self.attn_processors.keys())
        if isinstance(processor, dict):
            if len(processor) != len(names):
                raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not match the"
                                 f" number of attention layers: {len(names)}. Please make sure to pass {len(names)} processor classes.")
            processor = dict(processor)
        for name, mod in self.named_modules():
            if isinstance(mod, M.Attention):
                p = processor.pop(f"{name}.processor") if isinstance(processor, dict) else processor
                if not hasattr(p, "kind"):
                    raise TypeError(f"{type(p).__name__} is not an animate3d_amd processor (see animate3d_amd.modules)")
                mod.set_processor(p)
        self._invalidate()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        res = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._invalidate()
        return res

    def _apply(self, fn, *a, **k):
        self._invalidate()
        self._pe_cache = {}
        if getattr(self, "_ops_auto", False):       # .half() / .to(bfloat16) / .to(device): the op set follows the model
            self._ops, self._ops_auto = None, False
        return super()._apply(fn, *a, **k)

    @classmethod
    def from_unet2d(cls, unet, motion_adapter=None, load_weights: bool = True, **kw):
        """unet_motion_mv_model.py:275-368: build from a 2-D MVDream UNet (+ motion adapter) by copying
        state-dict entries; ``unet`` / ``motion_adapter`` only need ``state_dict()``."""
        model = cls(**kw)
        if load_weights:
            # exactly the members the reference copies (:332-360): conv_in, time / camera embedding, the resnets, attentions and
            # samplers of every block, conv_norm_out, conv_out — not encoder_hid_proj; motion modules come from the adapter (:394-402)
            own = re.compile(r"^(conv_in|time_embedding|camera_embedding|conv_norm_out|conv_out)\."
                             r"|^(down_blocks|up_blocks)\.\d+\.(resnets|attentions|downsamplers|upsamplers)\."
                             r"|^mid_block\.(resnets|attentions)\.")
            sd = {k: v for k, v in unet.state_dict().items() if own.match(k)}
            if motion_adapter is not None:
                sd.update({k: v for k, v in motion_adapter.state_dict().items()
                           if re.match(r"^((down_blocks|up_blocks)\.\d+|mid_block)\.motion_modules\.", k)})
            model.load_state_dict(sd, strict=False)
            # inference.py:161-165 initialises the I2V branch AFTER from_unet2d, from the PRETRAINED to_q (a plain 2-D UNet
            # carries no to_q_i2v): the construction-time copy took the random initialisation, so redo it for every
            # processor the source did not supply (a source that has the key wins, as in the reference's from_unet2d)
            with torch.no_grad():
                for name, blk, _c in model._blocks():
                    if blk.has_cross_attention:
                        for j, t in enumerate(blk.attentions):
                            tb = t.transformer_blocks[0]
                            key = f"{name}.attentions.{j}.transformer_blocks.0.attn1.processor.to_q_i2v.weight"
                            if getattr(tb.attn1.processor, "kind", None) == "mvdream_i2v" and key not in sd:
                                tb.attn1.processor.to_q_i2v.weight.copy_(tb.attn1.to_q.weight)
            model._invalidate()
        return model

    _MOTION_KEY = re.compile(r"^((down_blocks|up_blocks)\.\d+|mid_block)\.motion_modules\.")

    def load_motion_modules(self, motion_adapter) -> None:
        """unet_motion_mv_model.py:394-402: copy the motion modules of a MotionAdapter (anything with a ``state_dict()`` under the
        adapter's key names) into this model; an adapter without a mid block leaves the mid motion module as it is."""
        sd = {k: v for k, v in motion_adapter.state_dict().items() if self._MOTION_KEY.match(k)}
        own = self.state_dict()
        # the adapter carries the diffusers layers only; the processors' own parameters (to_*_sp, alpha_blender, ...) are not part of it
        want = [k for k in own if self._MOTION_KEY.match(k) and ".processor." not in k and (k in sd or not k.startswith("mid_block."))]
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError(f"motion adapter lacks {len(missing)} keys, e.g. {missing[0]}")
        self.load_state_dict(sd, strict=False)

    def save_motion_modules(self, save_directory: str, is_main_process: bool = True, safe_serialization: bool = True, **unused) -> None:
        """unet_motion_mv_model.py:404-438: write the motion modules in the layout of a diffusers MotionAdapter directory
        (``config.json`` + ``diffusion_pytorch_model.safetensors`` / ``.bin``): the diffusers layers of every motion module under the
        adapter's key names plus the sinusoidal ``pos_embed.pe`` buffers a stock ``MotionAdapter`` registers for its transformer blocks
        (this model computes them on the fly).  Round-tripped here through ``load_motion_modules``; loading the directory with
        ``MotionAdapter.from_pretrained`` could not be exercised offline (diffusers is not installed in this image).  The processors'
        own parameters (``to_*_sp``, ``alpha_blender`` ...) are not part of an adapter: they travel in the UNet checkpoint."""
        if not is_main_process:
            return
        import json
        cfg = self.config
        os.makedirs(save_directory, exist_ok=True)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items() if self._MOTION_KEY.match(k) and ".processor." not in k}
        for k in [k for k in sd if k.endswith(".transformer_blocks.0.attn1.to_q.weight")]:      # one buffer per temporal transformer block
            sd[k[:-len("attn1.to_q.weight")] + "pos_embed.pe"] = sinusoidal_pos_1d(sd[k].shape[0], cfg.motion_max_seq_length).contiguous()
        meta = {"_class_name": "MotionAdapter", "_diffusers_version": "0.27.2", "block_out_channels": list(cfg.block_out_channels),
                "motion_layers_per_block": cfg.layers_per_block, "motion_norm_num_groups": cfg.norm_num_groups,
                "motion_num_attention_heads": cfg.motion_num_attention_heads, "motion_max_seq_length": cfg.motion_max_seq_length,
                "use_motion_mid_block": True, "conv_in_channels": None}
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(meta, f, indent=2)
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "diffusion_pytorch_model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(save_directory, "diffusion_pytorch_model.bin"))

    def freeze_unet2d_params(self) -> None:
        """unet_motion_mv_model.py:370-392: freeze everything but the motion modules (fine-tuning AnimateDiff style; train.py itself
        selects by name, ``animate3d_amd.train.select_trainable``)."""
        for p in self.parameters():
            p.requires_grad = False
        for _name, blk, _c in self._blocks():
            for p in blk.motion_modules.parameters():
                p.requires_grad = True

    # diffusers conveniences of the reference class that change nothing on this implementation
    def fuse_qkv_projections(self):
        """:596-618.  The kernels always run fused projections (``_pack_t2d`` / ``_pack_motion``); nothing to do."""

    def unfuse_qkv_projections(self):
        """:620-631.  See ``fuse_qkv_projections``."""

    def enable_forward_chunking(self, chunk_size: Optional[int] = None, dim: int = 0) -> None:
        """:500-528 chunks the feed-forward to save memory; the fused GEGLU GEMM never materialises the 8C-wide projection, so
        there is nothing to chunk.  Accepted for call compatibility."""
        if dim not in (0, 1):
            raise ValueError(f"Make sure to set `dim` to either 0 or 1, not {dim}")

    def disable_forward_chunking(self) -> None:
        """:530-539."""

    def enable_freeu(self, s1: float, s2: float, b1: float, b2: float) -> None:
        """:562-585: the four FreeU factors (arXiv 2309.11497) are set on every up block; the arithmetic is diffusers' ``apply_freeu`` in front
        of each ``torch.cat([hidden_states, res_hidden_states], 1)`` of up blocks 0 and 1: the first half of the backbone channels x b1 / b2, the
        2 x 2 lowest frequencies of every skip plane x s1 / s2 (``_freeu``; inference only).  No caller of the reference uses it."""
        self._freeu_factors = (float(s1), float(s2), float(b1), float(b2))

    def disable_freeu(self) -> None:
        """:587-594."""
        self._freeu_factors = None

    def _freeu(self, idx: int, x, skip, B2: int, H: int, W: int):
        """``apply_freeu(resolution_idx = idx, ...)`` on token rows.  The backbone half is scaled in the storage type (as the reference's in-place
        multiply); the skip features go through rocFFT in fp32 over the (H, W) axes of their [B2, H, W, C] view (the reference transforms
        power-of-two planes in the model's dtype and others in fp32: fp32 always is the more accurate of the two) — like FreeInit's
        filter (denoise.py) this is torch.fft on device tensors, the one place besides it where the step leaves the C-ABI."""
        fz = getattr(self, "_freeu_factors", None)
        if fz is None or idx > 1:
            return x, skip
        s1, s2, b1, b2 = fz
        b, sc = (b1, s1) if idx == 0 else (b2, s2)
        half = x.shape[1] // 2
        x = x.clone()
        x[:, :half] *= b
        f = torch.fft.fftshift(torch.fft.fftn(skip.view(B2, H, W, -1).float(), dim=(1, 2)), dim=(1, 2))
        cr, cc = H // 2, W // 2
        f[:, cr - 1:cr + 1, cc - 1:cc + 1, :] *= sc
        skip = torch.fft.ifftn(torch.fft.ifftshift(f, dim=(1, 2)), dim=(1, 2)).real.to(skip.dtype).reshape(skip.shape).contiguous()
        return x, skip

    def set_default_attn_processor(self) -> None:
        """:542-555 only works while every processor is a stock diffusers one and raises otherwise; this model always carries the
        Animate3D processors, so it raises like the reference does in that state."""
        raise ValueError(f"Cannot call `set_default_attn_processor` when attention processors are of type {next(iter(self.attn_processors.values()))}")

    def _load_ip_adapter_weights(self, state_dict):
        """Counterpart of diffusers' UNet2DConditionLoadersMixin._load_ip_adapter_weights for the
        ip-adapter_sd15 layout {"image_proj": {proj.*, norm.*}, "ip_adapter": {"<2k+1>.to_k_ip.weight", ...}}
        (inference.py:78-85)."""
        if isinstance(state_dict, (list, tuple)):
            state_dict = state_dict[0]
        ip = self.encoder_hid_proj.image_projection_layers[0]
        img = state_dict["image_proj"]
        with torch.no_grad():
            ip.image_embeds.weight.copy_(img["proj.weight"]); ip.image_embeds.bias.copy_(img["proj.bias"])
            ip.norm.weight.copy_(img["norm.weight"]); ip.norm.bias.copy_(img["norm.bias"])
            key_id = 1
            for name, proc in self.attn_processors.items():
                if "motion_modules" in name:
                    continue
                if name.endswith("attn1.processor"):
                    continue
                proc.to_k_ip[0].weight.copy_(state_dict["ip_adapter"][f"{key_id}.to_k_ip.weight"])
                proc.to_v_ip[0].weight.copy_(state_dict["ip_adapter"][f"{key_id}.to_v_ip.weight"])
                key_id += 2
        self._invalidate()

    def init_synthetic(self, seed: int = 0):
        """Seeded on-device synthetic weights of realistic scale (no checkpoints exist offline):
        U(-1/sqrt(fan_in), 1/sqrt(fan_in)) matrices, norm gains near 1, small biases, every branch live
        (to_out_i2v / to_out_sp ~ N(0, 0.02), mix_factor ~ U(-0.5, 0.5))."""
        dev = self.device
        g = torch.Generator(device=dev).manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if p.ndim >= 2:
                    bound = 1.0 / math.sqrt(p[0].numel())
                    p.copy_((torch.rand(p.shape, generator=g, device=dev, dtype=torch.float32) * 2 - 1) * bound)
                elif name.endswith("mix_factor"):
                    p.copy_(torch.rand(p.shape, generator=g, device=dev) - 0.5)
                elif name.endswith("bias"):
                    p.copy_(0.02 * (torch.rand(p.shape, generator=g, device=dev, dtype=torch.float32) * 2 - 1))
                else:
                    p.copy_(1.0 + 0.1 * (torch.rand(p.shape, generator=g, device=dev, dtype=torch.float32) * 2 - 1))
            for name, p in self.named_parameters():
                if name.endswith("to_out_i2v.weight") or name.endswith("to_out_sp.weight"):
                    p.copy_(torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32) * 0.02)
        self._invalidate()
        return self

    # ------------------------------------------------------------------ op set / packed weights
    def _invalidate(self):
        """Weights / processors / dtype changed: every packed copy (and what the autograd op set derived from them) is stale."""
        self._packed = None
        self._packed_frozen = None
        if self._train_ops is not None:
            self._train_ops._persistent.clear()

    @property
    def ops(self):
        if self._active_ops is not None:       # a grad-enabled forward runs on the autograd view of the op set
            return self._active_ops
        return self._base_ops()

    def _base_ops(self):
        if self._ops is None:
            from .hip_ops import HipOps      # raises without an MI355X or without the built library
            # storage type of the kernels = the model's: fp16 for a .half() model (animatemv_guidance.py:339-346), bf16 otherwise
            self._ops = HipOps(self.device, torch.float16 if self.dtype == torch.float16 else torch.bfloat16)
            self._ops_auto = True
        return self._ops

    def _d(self, t: torch.Tensor) -> torch.Tensor:
        """Inference packs are detached copies; while ``_pack_train`` runs, a trainable parameter keeps its autograd history so
        that the gradient of the packed (cast / concatenated / interleaved) kernel operand flows back to it."""
        return t if (self._pack_grad and t.requires_grad) else t.detach()

    def _w(self, t: torch.Tensor) -> torch.Tensor:        # kernel weight: act dtype, contiguous
        if self._pack_grad and t.requires_grad and t.is_leaf:
            return self._wcat([t])
        return self._d(t).to(self.ops.act_dtype).contiguous()

    def _wcat(self, ws, interleave: bool = False) -> torch.Tensor:
        """Kernel weight of a fused projection: the rows of ``ws`` concatenated (GEGLU: interleaved in blocks of 32), act dtype.  Under
        ``_pack_train`` a trainable member makes it one ``autograd_ops.PackW`` node whose fp32 gradient comes straight from the GEMM backward."""
        if self._pack_grad and any(w.requires_grad for w in ws):
            from .autograd_ops import pack_weight
            return pack_weight(self.ops.act_dtype, list(ws), interleave)
        w = ws[0].detach() if len(ws) == 1 else torch.cat([w.detach() for w in ws], 0)
        if interleave:
            w = self.ops.interleave_geglu(w)
        return w.to(self.ops.act_dtype).contiguous()

    def _f(self, t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:   # bias / affine: fp32
        return None if t is None else self._d(t).float().contiguous()

    def _conv_w(self, conv: nn.Conv2d) -> torch.Tensor:   # [Cout, Cin, kh, kw] -> [Cout, kh*kw*Cin]
        w = self._d(conv.weight)
        return self._w(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1))

    def _fold(self, a: torch.Tensor, b_t: torch.Tensor) -> torch.Tensor:
        """fp32 ``a [M, K] @ b_t [N, K]^T`` on the HIP GEMM for pack-time weight folding: both operands are split into a 16-bit
        head and a 16-bit remainder (three products, fp32 accumulation), so the folded weight carries ~16 significant bits
        before it is rounded once to the storage type."""
        ops = self._base_ops()
        dt = ops.act_dtype
        a, b_t = a.detach().float().contiguous(), b_t.detach().float().contiguous()
        ah, bh = a.to(dt), b_t.to(dt)
        al, bl = (a - ah.float()).to(dt), (b_t - bh.float()).to(dt)
        return ops.gemm_f32out(ah, bh) + ops.gemm_f32out(ah, bl) + ops.gemm_f32out(al, bh)

    def _pack_resnet(self, r: M.ResnetBlock2D):
        return SimpleNamespace(
            n1=(self._f(r.norm1.weight), self._f(r.norm1.bias)), eps=r.norm1.eps,
            c1=(self._conv_w(r.conv1), self._f(r.conv1.bias)),
            temb=(self._w(r.time_emb_proj.weight), self._f(r.time_emb_proj.bias)),
            n2=(self._f(r.norm2.weight), self._f(r.norm2.bias)),
            c2=(self._conv_w(r.conv2), self._f(r.conv2.bias)),
            sc=None if r.conv_shortcut is None else (self._conv_w(r.conv_shortcut), self._f(r.conv_shortcut.bias)))

    def _pack_ff(self, tb):
        return SimpleNamespace(n3=(self._f(tb.norm3.weight), self._f(tb.norm3.bias)),
                               # GEGLU projection rows interleaved (h | gate in blocks of 32) for the fused GEMM epilogue
                               ff1=(self._wcat([tb.ff.net[0].proj.weight], interleave=True),
                                    self._f(self.ops.interleave_geglu(self._d(tb.ff.net[0].proj.bias)))),
                               ff2=(self._w(tb.ff.net[2].weight), self._f(tb.ff.net[2].bias)))

    def _pack_t2d(self, t: M.Transformer2DModel):
        tb = t.transformer_blocks[0]
        a1, a2 = tb.attn1, tb.attn2
        p1, p2 = a1.processor, a2.processor
        if p1.kind not in ("mvdream", "mvdream_i2v") or p2.kind != "ip_adapter":
            raise TypeError("Transformer2D layers need a MVDream(I2V) processor on attn1 and an IPAdapter processor on attn2")
        # fused projection rows ordered [K; V; Q; Q_i2v]: K|V and Q|Q_i2v are each one contiguous row range,
        # so the view-sharded path can run them as two GEMMs (K|V is what gets all-gathered)
        qkv = [a1.to_k.weight, a1.to_v.weight, a1.to_q.weight]
        i2v = p1.kind == "mvdream_i2v"
        if i2v:
            qkv.append(p1.to_q_i2v.weight)
        out = SimpleNamespace(
            heads=a1.heads, i2v=i2v,
            norm=(self._f(t.norm.weight), self._f(t.norm.bias)),
            pin=(self._conv_w(t.proj_in), self._f(t.proj_in.bias)),
            n1=(self._f(tb.norm1.weight), self._f(tb.norm1.bias)),
            qkv=self._wcat(qkv),
            oi2v=(self._w(p1.to_out_i2v.weight), self._f(p1.to_out_i2v.bias)) if i2v else None,
            o1=(self._w(a1.to_out[0].weight), self._f(a1.to_out[0].bias)),
            n2=(self._f(tb.norm2.weight), self._f(tb.norm2.bias)),
            q2=self._w(a2.to_q.weight),
            kv_text=self._wcat([a2.to_k.weight, a2.to_v.weight]),
            kv_ip=[self._wcat([k.weight, v.weight]) for k, v in zip(p2.to_k_ip, p2.to_v_ip)],
            ip_scale=list(p2.scale), ip_tokens=list(p2.num_tokens),
            o2=(self._w(a2.to_out[0].weight), self._f(a2.to_out[0].bias)),
            pout=(self._conv_w(t.proj_out), self._f(t.proj_out.bias)))
        out.o1m = None
        if i2v and not self._pack_grad:
            # to_out[0](main + to_out_i2v(i2v)) as ONE GEMM over [main | i2v] (attention_processor.py:375-383, 433-436): the weight is
            # [Wo | Wo Wi], the bias Wo b_i + b_o — one residual round trip instead of two HBM-bound N = K = C launches
            wo, wi = a1.to_out[0].weight.detach().float(), p1.to_out_i2v.weight.detach().float()
            bias = a1.to_out[0].bias.detach().float()
            if p1.to_out_i2v.bias is not None:
                bias = bias + self._fold(p1.to_out_i2v.bias.detach().float()[None, :], wo)[0]
            out.o1m = (self._w(torch.cat([wo, self._fold(wo, wi.t())], dim=1)), self._f(bias))
        out.ff = self._pack_ff(tb)
        return out

    def _pack_motion(self, m: M.TransformerTemporalModel):
        tb = m.transformer_blocks[0]
        attns = []
        for a, ln in ((tb.attn1, tb.norm1), (tb.attn2, tb.norm2)):
            pr = a.processor
            if pr.kind != "spatio_temporal":
                raise TypeError("motion-module attention layers need a SpatioTemporalI2VAttnProcessor")
            C = a.to_q.weight.shape[0]
            if hasattr(pr, "time_pos_embed"):
                pe = pr.time_pos_embed.pe[0]
            else:   # diffusers BasicTransformerBlock.pos_embed (sinusoidal) when the processor holds none
                pe = sinusoidal_pos_1d(C, self.config.motion_max_seq_length)[0].to(a.to_q.weight.device)
            proc_pe = hasattr(pr, "time_pos_embed")          # the processor restores the temporal encoding itself (:583-584)
            # merge weights: python floats for inference; under _pack_train the trainable mix_factor stays a tensor (its
            # gradient comes out of the to_out GEMMs' backward)
            ct, cs, ci = pr.blend_coefficients_t() if self._pack_grad else pr.blend_coefficients()
            if self._pack_grad:
                self._coef_tensors.extend(c for c in (ct, cs, ci) if torch.is_tensor(c))
            ns = SimpleNamespace(
                heads=a.heads, spatial=pr.use_spatial_attn, image=pr.use_image_attn,
                spatial_pe=pr.use_spatial_attn and pr.use_spatial_encoding, camera_pe=pr.use_spatial_attn and pr.use_camera_encoding,
                n=(self._f(ln.weight), self._f(ln.bias)),
                qkv=self._wcat([a.to_q.weight, a.to_k.weight, a.to_v.weight]),
                o=(self._w(a.to_out[0].weight), self._f(a.to_out[0].bias)),
                pe_t=self._w(pe), coef=(ct, cs, ci), proc=pr,
                # diffusers' BasicTransformerBlock.pos_embed stays on unless the spatial branch carries an encoding
                # (inference.py:176-178): the temporal PE then sits on the LayerNorm output that EVERY branch reads
                block_pe=not proc_pe)
            if pr.use_spatial_attn:
                ns.qkv_sp = self._wcat([pr.to_k_sp.weight, pr.to_v_sp.weight, pr.to_q_sp.weight])   # [K; V; Q]
                ns.osp = (self._w(pr.to_out_sp.weight), self._f(pr.to_out_sp.bias))
            if pr.use_image_attn:
                ns.qkv_img = self._wcat([pr.to_k_i2v.weight, pr.to_v_i2v.weight, pr.to_q_i2v.weight])
                ns.oimg = (self._w(pr.to_out_i2v.weight), self._f(pr.to_out_i2v.bias))
            ns.om = None
            if not self._pack_grad and (pr.use_spatial_attn or pr.use_image_attn):
                # ct to_out(t) + cs to_out_sp(sp) + ci to_out_i2v(img) (attention_processor.py:639, 666, 698-713) as ONE GEMM over
                # [t | sp | img] with the merge weights folded into [ct Wo | cs Wsp | ci Wimg]: one residual round trip instead of three
                parts = [(ct, a.to_out[0])]
                if pr.use_spatial_attn:
                    parts.append((cs, pr.to_out_sp))
                if pr.use_image_attn:
                    parts.append((ci, pr.to_out_i2v))
                bias = None
                for c, lin in parts:
                    if lin.bias is not None:
                        bias = float(c) * lin.bias.detach().float() if bias is None else bias + float(c) * lin.bias.detach().float()
                ns.om = (self._w(torch.cat([float(c) * lin.weight.detach().float() for c, lin in parts], dim=1)), self._f(bias))
            attns.append(ns)
        out = SimpleNamespace(norm=(self._f(m.norm.weight), self._f(m.norm.bias)),
                              pin=(self._w(m.proj_in.weight), self._f(m.proj_in.bias)),
                              attns=attns,
                              pout=(self._w(m.proj_out.weight), self._f(m.proj_out.bias)))
        out.ff = self._pack_ff(tb)
        return out

    def _pack_block(self, blk: M.MotionBlock):
        return SimpleNamespace(
            resnets=[self._pack_resnet(r) for r in blk.resnets],
            t2d=[self._pack_t2d(t) for t in blk.attentions] if blk.has_cross_attention else None,
            motion=[self._pack_motion(m) for m in blk.motion_modules],
            down=None if blk.downsamplers is None else (self._conv_w(blk.downsamplers[0].conv), self._f(blk.downsamplers[0].conv.bias)),
            up=None if blk.upsamplers is None else (self._conv_w(blk.upsamplers[0].conv), self._f(blk.upsamplers[0].conv.bias)))

    def _pack(self):
        """Kernel-layout copies of the weights (bf16 [N, K] matrices, fp32 biases/affines), built once."""
        cfg = self.config
        dev = self.device
        te, P = self.time_embedding, SimpleNamespace()
        P.time = (self._w(te.linear_1.weight), self._f(te.linear_1.bias), self._w(te.linear_2.weight), self._f(te.linear_2.bias))
        P.cam = None
        if cfg.camera_embedding_dim is not None:
            ce = self.camera_embedding
            w1 = torch.zeros(ce.linear_1.weight.shape[0], 64, device=dev, dtype=torch.float32)
            w1[:, : cfg.camera_embedding_dim] = ce.linear_1.weight.detach().float()
            P.cam = (self._w(w1), self._f(ce.linear_1.bias), self._w(ce.linear_2.weight), self._f(ce.linear_2.bias))
        P.ip = None
        if self.encoder_hid_proj is not None:
            ipl = self.encoder_hid_proj.image_projection_layers[0]
            P.ip = (self._w(ipl.image_embeds.weight), self._f(ipl.image_embeds.bias), self._f(ipl.norm.weight), self._f(ipl.norm.bias), ipl.norm.eps)
        # conv_in as a K=64 GEMM over im2col patches (k = (ky*3+kx)*Cin + ci)
        wi = self.conv_in.weight.detach().float().permute(0, 2, 3, 1).reshape(self.conv_in.weight.shape[0], -1)
        wpad = torch.zeros(wi.shape[0], 64, device=dev, dtype=torch.float32)
        wpad[:, : wi.shape[1]] = wi
        P.conv_in = (self._w(wpad), self._f(self.conv_in.bias))
        P.down = [self._pack_block(b) for b in self.down_blocks]
        P.mid = self._pack_block(self.mid_block)
        P.up = [self._pack_block(b) for b in self.up_blocks]
        P.norm_out = (self._f(self.conv_norm_out.weight), self._f(self.conv_norm_out.bias))
        P.conv_out = (self._conv_w(self.conv_out), self._f(self.conv_out.bias))
        P.cond = self._pack_conditioning(P)
        self._packed = P
        return P

    def _pack_conditioning(self, P):
        """Round 6.  The projections that see only the conditioning — ``time_emb_proj`` of the 22 ResNets (rows: SiLU(temb) per video),
        ``to_k | to_v`` of the 16 text cross-attentions and of their IP-adapter image tokens (unet_motion_mv_model.py:754-764 hands every
        block the same tokens) — stacked row-wise into three operands: three GEMMs per forward instead of 54 launches of one or a few tile
        rows each (15-30 us apiece, the latency of their K loops: ~1.3 ms of a 70-ms 4D-SDS step or rank step).  Every layer pack gets its
        row offset; the per-layer operands stay (training packs and the differentiable path use them)."""
        blocks = [*P.down, P.mid, *P.up]
        res = [r for b in blocks for r in b.resnets]
        t2d = [t for b in blocks if b.t2d is not None for t in b.t2d]
        c = SimpleNamespace(temb=None, kv_text=None, kv_ip=None, perm={}, widths=[r.temb[0].shape[0] for r in res])
        if res:
            off = 0
            for r in res:
                r.temb_off = (off, r.temb[0].shape[0])
                off += r.temb[0].shape[0]
            c.temb = (torch.cat([r.temb[0] for r in res], 0).contiguous(), torch.cat([r.temb[1] for r in res], 0).contiguous(), off)
        if t2d:
            off = 0
            for t in t2d:
                t.kv_off = (off, t.kv_text.shape[0])
                off += t.kv_text.shape[0]
            c.kv_text = torch.cat([t.kv_text for t in t2d], 0).contiguous()
            if all(len(t.kv_ip) == 1 and t.kv_ip[0].shape[0] == t.kv_text.shape[0] for t in t2d):
                c.kv_ip = torch.cat([t.kv_ip[0] for t in t2d], 0).contiguous()
        return c

    def _project_conditioning(self, c, semb, text_rows, ip_rows):
        """The three stacked projections of one forward (see _pack_conditioning).  The time projections are consumed as per-layer
        [rows, N] row-bias matrices of the first convolution (contiguous: a3d_conv3x3's rowbias has no row pitch), so the [rows, sum N]
        product is regrouped into consecutive [rows, N_i] blocks by one gather; the K | V products are read in place (column views)."""
        ops = self.ops
        out = SimpleNamespace(tp=None, M=0, kvt=None, kvi=None)
        if c.temb is not None:
            M = semb.shape[0]
            perm = c.perm.get(M)
            if perm is None:
                total = c.temb[2]
                idx = torch.arange(M * total, device=semb.device).view(M, total)
                parts, off = [], 0
                for w in c.widths:
                    parts.append(idx[:, off:off + w].reshape(-1))
                    off += w
                perm = c.perm[M] = torch.cat(parts).contiguous()
            out.tp = ops.gemm(semb, c.temb[0], c.temb[1]).reshape(-1).index_select(0, perm)
            out.M = M
        if c.kv_text is not None:
            out.kvt = ops.gemm(text_rows, c.kv_text)
            if c.kv_ip is not None and len(ip_rows) == 1:
                out.kvi = ops.gemm(ip_rows[0], c.kv_ip)
        return out


    def _pe_spatial(self, C: int, h: int, w: int) -> torch.Tensor:
        """2-D sine PE table [h*w, C] for one level (embeddings.py:59-96), cached per geometry."""
        key = (C, h, w)
        if key not in self._pe_cache:
            self._pe_cache[key] = sine_pos_2d(C // 2, h, w).to(device=self.device, dtype=self.ops.act_dtype).contiguous()
        return self._pe_cache[key]

    def _pe_spatial_general(self, a, C: int, h: int, w: int, n: int, F: int, view0: int = 0):
        """(table, rows-per-entry divisor) added to the spatial branch's tokens: the 2-D encoding (sinusoid or the processor's
        learnable row / column tables, embeddings.py:52-157) plus, with camera encoding on, one vector per view
        (attention_processor.py:565-575).  Without camera encoding the table is [h*w, C] (index = token within the image);
        with it the index runs over (view, frame, token) of one batch element, [n*F*h*w, C].  ``view0``: first global view
        of a view shard."""
        pr = a.proc
        L = h * w
        pe = None
        if self._active_ops is not None:       # a differentiable forward: the tables below enter as constants
            learn = ([pr.spatial_pos_embed.row_embed.weight, pr.spatial_pos_embed.col_embed.weight]
                     if a.spatial_pe and pr.spatial_encoding_type == "learnable" else [])
            learn += [pr.camera_embed.embedding_table.weight] if a.camera_pe and pr.camera_encoding_type == "learnable" else []
            if any(t.requires_grad for t in learn):
                raise NotImplementedError("learnable positional / camera-encoding tables are not trainable on this path (the reference's "
                                          "training configuration uses the sinusoid encodings); freeze them (requires_grad_(False))")
        if a.spatial_pe:
            if pr.spatial_encoding_type == "learnable":
                sp = pr.spatial_pos_embed
                if h > sp.row_embed.weight.shape[0] or w > sp.col_embed.weight.shape[0]:
                    raise ValueError(f"feature map {(h, w)} exceeds the learnable positional tables "
                                     f"({sp.row_embed.weight.shape[0]} x {sp.col_embed.weight.shape[0]})")
                xe = sp.col_embed.weight.detach().float()[:w][None, :, :].expand(h, w, -1)
                ye = sp.row_embed.weight.detach().float()[:h][:, None, :].expand(h, w, -1)
                pe = torch.cat([xe, ye], dim=-1).reshape(L, C).to(self.device)
            else:
                pe = sine_pos_2d(C // 2, h, w).to(self.device)
        if not a.camera_pe:
            return pe.to(self.ops.act_dtype).contiguous(), 1
        cam = pr.camera_embed.embedding_table.weight if pr.camera_encoding_type == "learnable" else pr.camera_embed.pe[0]
        cam = cam.detach().float().to(self.device)[view0:view0 + n]
        if cam.shape[0] != n:
            raise ValueError(f"camera encoding holds {pr.camera_embed} vectors, the call has views {view0}..{view0 + n}")
        tab = cam[:, None, None, :].expand(n, F, L, C)
        if pe is not None:
            tab = tab + pe[None, None, :, :]
        return tab.reshape(n * F * L, C).to(self.ops.act_dtype).contiguous(), 1

    # ------------------------------------------------------------------ forward pieces
    def _resnet(self, x, B2, H, W, pk, semb, rb_rows):
        """``x``: token rows, or (up blocks, inference) the pair (hidden, skip) whose channel concatenation the reference feeds in
        (unet_motion_mv_model.py:826-827 / diffusers ``torch.cat([hidden_states, res_hidden_states], dim=1)``): norm1 and the 1x1 shortcut
        read the two parts in place (a3d_group_norm2 / a3d_gemm2), the concatenated tensor is never written."""
        ops, g = self.ops, self.config.norm_num_groups
        L = H * W
        sc = None
        if isinstance(x, tuple):
            xa, xb = x
            sc = ops.gemm2(xa, xb, pk.sc[0], pk.sc[1]) if pk.sc is not None else None
            if sc is None:                      # shape outside the persistent kernel (or no shortcut conv): concatenate after all
                x = ops.concat(xa, xb)
                h = ops.group_norm(x, B2, L, pk.n1[0], pk.n1[1], g, pk.eps, True)
            else:
                h = ops.group_norm2(xa, xb, B2, L, pk.n1[0], pk.n1[1], g, pk.eps, True)
        else:
            h = ops.group_norm(x, B2, L, pk.n1[0], pk.n1[1], g, pk.eps, True)
        c = self._cond
        if c is not None and c.tp is not None and hasattr(pk, "temb_off"):       # inference: one stacked projection per forward
            o, N = pk.temb_off
            tp = c.tp[c.M * o: c.M * (o + N)].view(c.M, N)
        else:
            tp = ops.gemm(semb, pk.temb[0], pk.temb[1])                   # time_emb_proj(SiLU(temb))
        h, _, _ = ops.conv3x3(h, B2, H, W, pk.c1[0], pk.c1[1], rowbias=tp, rb_div=rb_rows * L)
        h = ops.group_norm(h, B2, L, pk.n2[0], pk.n2[1], g, pk.eps, True)
        if sc is None:
            sc = x if pk.sc is None else ops.gemm(x, pk.sc[0], pk.sc[1])
        out, _, _ = ops.conv3x3(h, B2, H, W, pk.c2[0], pk.c2[1], residual=sc)
        return out

    def _ff(self, h, pk):
        ops = self.ops
        n3 = ops.layer_norm(h, pk.n3[0], pk.n3[1], 1e-5)
        u = ops.gemm_geglu(n3, pk.ff1[0], pk.ff1[1])           # proj + h * gelu(gate) in one kernel
        return ops.gemm(u, pk.ff2[0], pk.ff2[1], residual=h)

    def _mv_maps(self, n, F, L):
        qm = RowMap(gdiv=F, ga=n * F * L, gb=L, seg_len=L, seg_stride=F * L)     # "(b n f) l -> (b f) (n l)"
        k0 = RowMap(gdiv=F, ga=n * F * L, gb=0, seg_len=L, seg_stride=F * L)     # same, frame 0 of every b
        return qm, k0

    def _mv_attention(self, x, w_kvq, C, V, n, F, L, heads, i2v, overlap=None, out_a=None, out_ai=None):
        """Multi-view attention over the n*L tokens of every (b, f) group (+ the first-frame branch).
        ``w_kvq`` rows are [K; V; Q; (Q_i2v)].  Unsharded: one fused GEMM, K/V/Q are column views.
        View-sharded (animate3d_amd.parallel): this rank holds n of the N views; K|V is projected
        into its own contiguous buffer, all-gathered over the view group (RCCL) and the kernels
        read the gathered K/V through the unsharded row map while Q stays local.  The gather is
        asynchronous: the Q projection and ``overlap()`` (independent work of the caller, e.g. the
        temporal branch of a motion module) are issued while it is in flight.  ``out_a`` / ``out_ai``: [rows, C] views (own row
        stride) the two attention outputs are written to — the column blocks of a merged out-projection's A operand.
        Returns (a, a_i2v, overlap())."""
        ops, par = self.ops, self.parallel
        if par is not None and par.world == 1:
            par = None
        qm, k0 = self._mv_maps(n, F, L)
        b = V // n
        fsh = par is not None and par.frame_shards > 1
        vsh = par is not None and par.view_shards > 1
        N = n * par.view_shards if vsh else n

        def first_frame_kv():
            """K|V of frame 0 of every b for the I2V branch when this rank may not hold frame 0: its tokens come from the rank
            of the frame group that does, then (view-sharded) from the other view shards; [b * N * L, 2C], rows (b N) l."""
            x0 = x.view(V, F, L, C)[:, 0].reshape(V * L, C) if par.frame_rank == 0 else None
            x0 = par.broadcast_frame0(x0, (V * L, C), x.dtype, x.device)
            if vsh:
                x0 = par.all_gather_views(x0, b)
            return ops.gemm(x0, _rows(w_kvq, 0, 2 * C)), RowMap(gdiv=F, ga=N * L, gb=0, seg_len=L, seg_stride=L)

        if not vsh:
            kvq = ops.gemm(x, w_kvq)
            k, v, q, *qi = ops.split_cols(kvq, *range(0, kvq.shape[1] + 1, C))
            oa = {} if out_a is None else {"out": out_a}
            oi = {} if out_ai is None else {"out": out_ai}
            a = ops.flash_attn(q, k, v, qm, qm, b * F, heads, n * L, n * L, **oa)
            ai = None
            if i2v and fsh:
                kv0, km0 = first_frame_kv()
                ai = ops.flash_attn(qi[0], kv0[:, :C], kv0[:, C:], qm, km0, b * F, heads, n * L, n * L, **oi)
            elif i2v:
                ai = ops.flash_attn(qi[0], k, v, qm, k0, b * F, heads, n * L, n * L, **oi)
            return a, ai, (overlap() if overlap is not None else None)
        if par.gather_tokens:
            # gather the (normalised) INPUT tokens [rows_local, C] and project K|V for all N views locally: half the
            # bytes on xGMI for S x the (cheap, HBM-bound) K|V projection
            pending = par.all_gather_views_start(x, b)
            qq = ops.gemm(x, _rows(w_kvq, 2 * C))                   # overlaps the gather
            extra = overlap() if overlap is not None else None
            kv_all = ops.gemm(par.all_gather_views_finish(pending, b), _rows(w_kvq, 0, 2 * C))
        else:
            kv = ops.gemm(x, _rows(w_kvq, 0, 2 * C))                   # [rows_local, 2C], contiguous
            pending = par.all_gather_views_start(kv, b)       # RCCL stream
            qq = ops.gemm(x, _rows(w_kvq, 2 * C))                   # [rows_local, C or 2C], overlaps the gather
            extra = overlap() if overlap is not None else None
            kv_all = par.all_gather_views_finish(pending, b)  # [b * N*F*L, 2C] in unsharded (b n f) l order
        km, km0 = self._mv_maps(N, F, L)
        k, v = kv_all[:, :C], kv_all[:, C:]
        oa = {} if out_a is None else {"out": out_a}
        oi = {} if out_ai is None else {"out": out_ai}
        a = ops.flash_attn(qq[:, :C], k, v, qm, km, b * F, heads, n * L, N * L, **oa)
        ai = None
        if i2v and fsh:
            kv0, km0 = first_frame_kv()
            ai = ops.flash_attn(qq[:, C:2 * C], kv0[:, :C], kv0[:, C:], qm, km0, b * F, heads, n * L, N * L, **oi)
        elif i2v:
            ai = ops.flash_attn(qq[:, C:2 * C], k, v, qm, km0, b * F, heads, n * L, N * L, **oi)
        return a, ai, extra

    def _t2d(self, x, V, n, F, H, W, pk, text_rows, ip_rows, T):
        ops, g = self.ops, self.config.norm_num_groups
        B2, L, C = V * F, H * W, x.shape[1]
        h = ops.group_norm(x, B2, L, pk.norm[0], pk.norm[1], g, 1e-6, False)
        h = ops.gemm(h, pk.pin[0], pk.pin[1])
        # attn1: multi-view self-attention (+ first-frame attention)
        n1 = ops.layer_norm(h, pk.n1[0], pk.n1[1], 1e-5)
        h = self._self_attention(n1, h, pk, V, n, F, L)
        # attn2: text + IP-Adapter cross-attention, K/V projected once per video
        n2 = ops.layer_norm(h, pk.n2[0], pk.n2[1], 1e-5)
        h = self._cross_attention(n2, h, pk, text_rows, ip_rows, T, B2, F, L)
        h = self._ff(h, pk.ff)
        return ops.gemm(h, pk.pout[0], pk.pout[1], residual=x)

    def _cross_attention(self, n2, residual, pk, text_rows, ip_rows, T, B2, F, L):
        """IPAdapter processor (attention_processor.py:169-298) on normalised tokens ``n2``: text attention plus, per adapter,
        ``scale`` x image-token attention accumulated into the same buffer, then ``to_out`` (+ ``residual`` if given)."""
        ops = self.ops
        C = n2.shape[1]
        q2 = ops.gemm(n2, pk.q2)
        c = self._cond if hasattr(pk, "kv_off") else None
        kvt = c.kvt[:, pk.kv_off[0]: pk.kv_off[0] + 2 * C] if c is not None and c.kvt is not None else ops.gemm(text_rows, pk.kv_text)
        qc = RowMap(gdiv=1, ga=L, gb=0, seg_len=L, seg_stride=0)
        ca = None
        fused = getattr(ops, "flash_attn2", None)
        if fused is not None and len(ip_rows) == 1:          # one adapter (the released configuration): text + image tokens in one launch
            kvi = c.kvi[:, pk.kv_off[0]: pk.kv_off[0] + 2 * C] if c is not None and c.kvi is not None else ops.gemm(ip_rows[0], pk.kv_ip[0])
            nt = pk.ip_tokens[0]
            ca = fused(q2, kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], qc, RowMap(F, T, 0, T, 0), RowMap(F, nt, 0, nt, 0), B2, pk.heads,
                       L, T, nt, out_scale2=pk.ip_scale[0])
        if ca is None:
            ca = ops.flash_attn(q2, kvt[:, :C], kvt[:, C:], qc, RowMap(F, T, 0, T, 0), B2, pk.heads, L, T, accumulation_target=bool(ip_rows))
            for ipr, w, scale, nt in zip(ip_rows, pk.kv_ip, pk.ip_scale, pk.ip_tokens):
                kvi = ops.gemm(ipr, w)
                ops.flash_attn(q2, kvi[:, :C], kvi[:, C:], qc, RowMap(F, nt, 0, nt, 0), B2, pk.heads, L, nt,
                               out=ca, out_scale=scale, accumulate=True)
        return ops.gemm(ca, pk.o2[0], pk.o2[1], residual=residual)

    def _self_attention(self, n1, residual, pk, V, n, F, L):
        """MVDream(I2V) processor (attention_processor.py:39-126, 325-445) on normalised tokens ``n1``."""
        ops = self.ops
        C = n1.shape[1]
        if pk.o1m is not None and self._active_ops is None:
            # inference: both attention kernels write into the column halves of one [rows, 2C] buffer, one K = 2C out-projection
            cat = ops.empty(n1.shape[0], 2 * C)
            self._mv_attention(n1, pk.qkv, C, V, n, F, L, pk.heads, i2v=True, out_a=cat[:, :C], out_ai=cat[:, C:])
            return ops.gemm(cat, pk.o1m[0], pk.o1m[1], residual=residual)
        a, ai, _ = self._mv_attention(n1, pk.qkv, C, V, n, F, L, pk.heads, i2v=pk.i2v)
        if pk.i2v:
            a = ops.gemm(ai, pk.oi2v[0], pk.oi2v[1], residual=a)          # main + to_out_i2v(i2v)
        return ops.gemm(a, pk.o1[0], pk.o1[1], residual=residual)

    def _motion_attn(self, h, nt, ns, nimg, a, V, n, F, L):
        """SpatioTemporalI2V processor (attention_processor.py:541-723) + the block's residual: ``nt`` / ``ns`` / ``nimg`` are the
        inputs of the temporal, spatial and first-frame-image branches (normalised tokens with their encodings); returns
        ``h + ct * temporal + cs * spatial + ci * image`` with the merge coefficients of the processor."""
        ops = self.ops
        C = nt.shape[1]
        ct, cs, ci = a.coef

        par = self.parallel if (self.parallel is not None and self.parallel.world > 1) else None
        fsh = par is not None and par.frame_shards > 1
        F_all, f0 = self._frames
        # inference: the branch kernels write the column blocks [t | sp | img] of ONE buffer and a single out-projection with the
        # merge weights folded in (``a.om``) replaces the two / three residual-accumulating ones
        merged = a.om is not None and self._active_ops is None
        nblk = 1 + int(a.spatial) + int(a.image)
        cat = ops.empty(nt.shape[0], nblk * C) if merged else None
        blk = lambda i: {} if cat is None else {"out": cat[:, i * C:(i + 1) * C]}

        def temporal_branch(nt=nt, a=a):
            if not fsh:
                qkv = ops.gemm(nt, a.qkv)
                return ops.temporal_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], V, F, L, a.heads, **blk(0))
            # frame-sharded: every rank needs the keys / values of ALL frames of its pixels; queries stay local
            kv = ops.gemm(nt, _rows(a.qkv, C))                      # [rows_local, 2C], contiguous
            pending = par.all_gather_frames_start(kv)
            q = ops.gemm(nt, _rows(a.qkv, 0, C))                       # overlaps the gather
            kv_all = par.all_gather_frames_finish(pending)    # [Sf, (v f_l) l, 2C], read in place
            return ops.temporal_attn(q, kv_all[:, :C], kv_all[:, C:], V, F_all, L, a.heads, q_f0=f0, q_frames=F, **blk(0))
        if a.spatial:
            # the temporal branch is independent of the multi-view one: it runs while the K|V gather is in flight
            asp, _, at = self._mv_attention(ns, a.qkv_sp, C, V, n, F, L, a.heads, i2v=False, overlap=temporal_branch,
                                            out_a=None if cat is None else cat[:, C:2 * C])
        else:
            at = temporal_branch()
        if not merged:
            out = ops.gemm(at, a.o[0], a.o[1], residual=h, alpha=ct, beta=1.0)
            if a.spatial:
                out = ops.gemm(asp, a.osp[0], a.osp[1], residual=out, alpha=cs, beta=1.0)
        if a.image:
            # per-view first-frame attention (:672-698): every image attends to frame 0 of its own video
            qm = RowMap(gdiv=1, ga=L, gb=0, seg_len=L, seg_stride=0)
            if fsh:                       # frame 0's tokens from the rank that holds them
                x0 = nimg.view(V, F, L, C)[:, 0].reshape(V * L, C) if par.frame_rank == 0 else None
                kv0 = ops.gemm(par.broadcast_frame0(x0, (V * L, C), nimg.dtype, nimg.device), _rows(a.qkv_img, 0, 2 * C))
                qi = ops.gemm(nimg, _rows(a.qkv_img, 2 * C))
                ai = ops.flash_attn(qi, kv0[:, :C], kv0[:, C:], qm, RowMap(gdiv=F, ga=L, gb=0, seg_len=L, seg_stride=0), V * F, a.heads, L, L,
                                    **blk(nblk - 1))
            else:
                kvq = ops.gemm(nimg, a.qkv_img)
                k0 = RowMap(gdiv=F, ga=F * L, gb=0, seg_len=L, seg_stride=0)
                ki, vi, qi = ops.split_cols(kvq, 0, C, 2 * C, 3 * C)
                ai = ops.flash_attn(qi, ki, vi, qm, k0, V * F, a.heads, L, L, **blk(nblk - 1))
            if not merged:
                out = ops.gemm(ai, a.oimg[0], a.oimg[1], residual=out, alpha=ci, beta=1.0)
        if merged:
            out = ops.gemm(cat, a.om[0], a.om[1], residual=h)
        return out

    def _motion(self, x, V, n, F, H, W, pk):
        ops, g = self.ops, self.config.norm_num_groups
        L, C = H * W, x.shape[1]
        par = self.parallel if (self.parallel is not None and self.parallel.world > 1) else None
        F_all, f0 = self._frames
        if par is not None and par.frame_shards > 1:
            # 3-D GroupNorm per video over ALL frames: fp64 partial sums of the local frames, one all-reduce over the frame group
            sums = par.all_reduce_frames(ops.group_norm_sums(x, V, F * L, g))
            cnt = float(F_all * L * (C // g))
            mean = sums[..., 0] / cnt
            var = (sums[..., 1] / cnt - mean * mean).clamp_min(0.0)
            stats = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-6)], dim=-1).to(torch.float32).contiguous()
            h = ops.group_norm_apply(x, V, F * L, pk.norm[0], pk.norm[1], g, stats, False)
        else:
            h = ops.group_norm(x, V, F * L, pk.norm[0], pk.norm[1], g, 1e-6, False)     # 3-D GroupNorm per video
        h = ops.gemm(h, pk.pin[0], pk.pin[1])
        view0 = par.view_rank * n if par is not None else 0
        for a in pk.attns:
            pe_t = a.pe_t[f0:f0 + F]
            ln = lambda **kw: ops.layer_norm(h, a.n[0], a.n[1], 1e-5, **kw)
            if a.block_pe:                     # the block adds the temporal encoding to what every branch reads
                nt = ns = nimg = ln(pe1=pe_t, pe1_div=L)
            else:                              # the processor adds it to the temporal branch only (:583-584)
                if a.spatial_pe and not a.camera_pe and a.proc.spatial_encoding_type == "sinusoid":
                    pe_s, div_s = self._pe_spatial(C, H, W), 1
                else:
                    pe_s, div_s = self._pe_spatial_general(a, C, H, W, n, F, view0)
                nt, ns = ln(pe1=pe_t, pe1_div=L, pe2=pe_s, pe2_div=div_s, two=True)
                nimg = ln() if a.image else None
            h = self._motion_attn(h, nt, ns, nimg, a, V, n, F, L)
        h = self._ff(h, pk.ff)
        return ops.gemm(h, pk.pout[0], pk.pout[1], residual=x)

    # ------------------------------------------------------------------ HIP-graph replay
    def capture_graph(self, **inputs):
        """Capture one forward into a HIP graph and return ``step(**new_inputs) -> UNet3DConditionOutput``: every call copies
        the new tensors into the captured input buffers and replays the ~1 700 launches without host work.  Measured on MI355X the
        step is GPU-bound even at the small BASELINE sizes (config 5: 85.9 ms eager, 86.1 ms replay; config 1: 34.1 / 34.0 ms),
        so this only helps when the host thread is busy elsewhere.  ``inputs`` are the ``forward`` keyword
        arguments (plus ``sample`` / ``timestep``); non-tensor arguments are baked in.  The returned ``.sample`` is a static
        buffer that the next call overwrites.  Possible because the C-ABI never allocates, never synchronises and launches only
        on the caller's current stream (include/animate3d_hip.h)."""
        clone = lambda v: v.clone() if torch.is_tensor(v) else ({k: w.clone() for k, w in v.items()} if isinstance(v, dict) else v)
        static = {k: clone(v) for k, v in inputs.items()}
        self(**static)                                   # weight packing + one-time kernel attribute calls outside the capture
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self(**static)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self(**static)

        def step(**new_inputs):
            for k, v in new_inputs.items():
                if torch.is_tensor(v):
                    static[k].copy_(v)
                elif isinstance(v, dict):
                    for kk, vv in v.items():
                        static[k][kk].copy_(vv)
                elif v != static[k]:
                    raise ValueError(f"capture_graph: non-tensor argument {k!r} is baked into the graph")
            graph.replay()
            return out
        return step

    # ------------------------------------------------------------------ training
    def enable_training(self, compute_dtype: Optional[torch.dtype] = None):
        """Make grad-enabled forwards differentiable (train.py:576-590: ``unet(...)`` then ``loss.backward()``): they run on
        ``animate3d_amd.autograd_ops.AutogradOps`` — the same kernels forward, the backward kernels of the C-ABI's training section
        behind ``torch.autograd.Function`` — and read the trainable parameters (``requires_grad``; the reference trains
        ``motion_modules.`` and ``i2v.``, configs/training/train.yaml) through per-step packed copies that keep their autograd
        history, so ``.grad`` lands on the fp32 ``nn.Parameter`` exactly as under the reference's autocast.  ``compute_dtype``
        (bf16 default, or fp16 = the reference's autocast type; then use a loss scaler) is the kernels' storage type for a model
        kept in fp32.  ``torch.no_grad()`` forwards are unchanged.  Returns ``self``."""
        if compute_dtype is not None and self._base_ops().act_dtype != compute_dtype:
            if not getattr(self, "_ops_auto", False):
                raise ValueError("enable_training(compute_dtype=...) cannot replace an op set that was passed in")
            from .hip_ops import HipOps
            self._ops = HipOps(self.device, compute_dtype)
            self._invalidate()
        self._train_dtype = compute_dtype
        self._training_enabled = True
        self._autograd_ops()
        return self

    def _autograd_ops(self):
        """The autograd view of the CURRENT op set (``unet.to(device)`` after ``enable_training()`` — train.py:456 — replaces it)."""
        from .autograd_ops import AutogradOps
        base = self._base_ops()
        want = getattr(self, "_train_dtype", None)
        if want is not None and base.act_dtype != want and getattr(self, "_ops_auto", False):
            from .hip_ops import HipOps
            self._ops = base = HipOps(self.device, want)
            self._invalidate()
        if self._train_ops is None or self._train_ops.base is not base:
            self._train_ops = AutogradOps(base)
            self._packed_frozen = None
        return self._train_ops

    def enable_gradient_checkpointing(self):
        """train.py:381-382 (diffusers wraps every ResNet / Transformer2D / motion module in ``torch.utils.checkpoint``).  Here one
        checkpoint spans a whole layer (ResNet + Transformer2D + motion module): a grad-enabled forward keeps only the layer boundaries and
        recomputes the inside of a layer when the backward reaches it — the HIP kernels run again through the same autograd functions, so
        the gradients are bit-identical to the un-checkpointed ones.  The train.yaml shape needs 38 GB without it (of 288 GB per MI355X),
        so the reference's default costs one extra forward here for nothing; larger batches / resolutions are where it pays."""
        self._gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        self._gradient_checkpointing = False

    def _mark_persistent(self, node):
        if torch.is_tensor(node):
            node._a3d_persistent = True
        elif isinstance(node, SimpleNamespace):
            for v in vars(node).values():
                self._mark_persistent(v)
        elif isinstance(node, (list, tuple)):
            for v in node:
                self._mark_persistent(v)

    def _pack_train(self):
        """Pack for one differentiable forward: sub-modules without a trainable parameter come from a persistent detached pack
        (so that the autograd op set can cache their transposed / flipped dgrad operands), the others are packed again with
        ``_pack_grad`` on — a cast / cat / interleave per step, the price of fp32 master weights behind 16-bit kernels."""
        # the frozen pack is valid for ONE set of trainable parameters: requires_grad_() / select_trainable() / freeze_unet2d_params()
        # between steps (staged training: a module trained first and frozen later) must not leave its pre-training weights in the pack
        sig = tuple(p.requires_grad for p in self.parameters())
        if self._packed_frozen is None or sig != getattr(self, "_packed_frozen_sig", None):
            self._packed_frozen = self._pack()
            self._packed_frozen_sig = sig
            self._packed = None                    # (_pack stored it as the inference pack; that one is rebuilt on demand)
            self._mark_persistent(self._packed_frozen)
            if getattr(self, "_train_ops", None) is not None:
                self._train_ops._persistent.clear()    # (weight, derived dgrad operand) pairs of the previous frozen pack: several GB per switch
        Pf = self._packed_frozen
        trainable = lambda m: m is not None and any(p.requires_grad for p in m.parameters())
        for name, mod in (("time_embedding", self.time_embedding), ("camera_embedding", getattr(self, "camera_embedding", None)),
                          ("encoder_hid_proj", self.encoder_hid_proj), ("conv_in", self.conv_in), ("conv_norm_out", self.conv_norm_out),
                          ("conv_out", self.conv_out)):
            if trainable(mod):
                raise NotImplementedError(f"{name} is not trainable on this path (the reference trains 'motion_modules.' and 'i2v.' only)")
        self._pack_grad = True
        self._coef_tensors = []
        try:
            def block(blk, pf):
                out = SimpleNamespace(**vars(pf))
                out.resnets = [self._pack_resnet(r) if trainable(r) else q for r, q in zip(blk.resnets, pf.resnets)]
                if pf.t2d is not None:
                    out.t2d = [self._pack_t2d(t) if trainable(t) else q for t, q in zip(blk.attentions, pf.t2d)]
                out.motion = [self._pack_motion(m) if trainable(m) else q for m, q in zip(blk.motion_modules, pf.motion)]
                if (blk.downsamplers is not None and trainable(blk.downsamplers[0])) or (blk.upsamplers is not None and trainable(blk.upsamplers[0])):
                    raise NotImplementedError("trainable down / up-sampler convolutions are not supported")
                return out
            P = SimpleNamespace(**vars(Pf))
            P.down = [block(b, q) for b, q in zip(self.down_blocks, Pf.down)]
            P.mid = block(self.mid_block, Pf.mid)
            P.up = [block(b, q) for b, q in zip(self.up_blocks, Pf.up)]
        finally:
            self._pack_grad = False
        if self._train_ops is not None:
            self._train_ops.prefetch_scalars(self._coef_tensors)      # every trainable merge weight in one device -> host transfer
        return P

    # ------------------------------------------------------------------ forward
    @on_model_device
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int], encoder_hidden_states: torch.Tensor,
                timestep_cond: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                cross_attention_kwargs: Optional[Dict[str, Any]] = None, added_cond_kwargs: Optional[Dict[str, torch.Tensor]] = None,
                down_block_additional_residuals=None, mid_block_additional_residual=None, return_dict: bool = True,
                camera: Optional[torch.Tensor] = None, num_views: int = 4, i2v_cond_time_zero: bool = False):
        """Same contract as the reference forward (unet_motion_mv_model.py:633-867): ``sample``
        [V, C, F, h, w] with V = b*cfg*views in (b n) order, returns ``.sample`` of the same shape.
        Inference (the reference's callers wrap it in no_grad: pipeline.py:758, guidance :422) runs without autograd; after
        ``enable_training()`` a grad-enabled call is differentiable (train.py:576-590)."""
        kw = dict(timestep_cond=timestep_cond, attention_mask=attention_mask, cross_attention_kwargs=cross_attention_kwargs,
                  added_cond_kwargs=added_cond_kwargs, down_block_additional_residuals=down_block_additional_residuals,
                  mid_block_additional_residual=mid_block_additional_residual, return_dict=return_dict, camera=camera,
                  num_views=num_views, i2v_cond_time_zero=i2v_cond_time_zero)
        if getattr(self, "_training_enabled", False) and torch.is_grad_enabled():
            if self.parallel is not None and self.parallel.world > 1:
                raise NotImplementedError("a sharded (shard_unet) model is inference-only; training shards the batch (train.py: DDP)")
            if getattr(self, "_freeu_factors", None) is not None:
                raise NotImplementedError("FreeU (enable_freeu) is an inference-time re-weighting; the training path does not differentiate it")
            self._active_ops = self._autograd_ops()
            self._packed = None        # the parameters are about to change: the next no_grad call (validation) packs them afresh
            try:
                return self._forward_impl(sample, timestep, encoder_hidden_states, packed=self._pack_train(), **kw)
            finally:
                self._active_ops = None
        with torch.no_grad():
            try:
                return self._forward_impl(sample, timestep, encoder_hidden_states, packed=None, **kw)
            finally:
                self._cond = None          # (layer methods called on their own, e.g. by the per-block tests, project per layer)

    def _reject_attention_mask(self, attention_mask, sample, num_views):
        """``attention_mask`` (:639, 700-703, 778-841) cannot be honoured — by the reference either.  It becomes an additive bias
        [B, 1, K] that every Transformer2D ``attn1`` hands to xformers after ``(b n f) l c -> (b f) (n l) c`` (attention_processor.py:340,
        361-370, 405, 416); diffusers' ``prepare_attention_mask`` pads a mask whose length is not n * l BY n * l, so the bias can match
        xformers' required [b F heads, n l, n l] at ONE resolution at most, and this UNet attends at several: the reference raises inside
        xformers for any mask.  Same here, with the shape walk in the message (oracle/unet_ref.py: reference_attention_mask_trace restates
        it; tests/test_host_logic.py pins it).  A UNet with attention at a single resolution would be consistent for a [b F, n l] mask;
        that bias is not implemented in the attention kernels."""
        if attention_mask.dim() != 2:
            raise ValueError(f"attention_mask must be [batch, key_tokens] (unet_motion_mv_model.py:662), got {tuple(attention_mask.shape)}")
        V, n, F = sample.shape[0], num_views, sample.shape[2]
        h, w = sample.shape[-2:]
        ls, hh, ww = [], h, w
        for i in range(len(self.config.block_out_channels)):
            if self.config.down_has_attn[i] or i == len(self.config.block_out_channels) - 1:
                ls.append(hh * ww)
            if i < len(self.config.block_out_channels) - 1:
                hh, ww = (hh + 1) // 2, (ww + 1) // 2
        B, K = attention_mask.shape
        heads, b = self.config.num_attention_heads, V // n
        bad = []
        for l in ls:
            k_eff = K if K == n * l else K + n * l
            b_eff = B * heads if B < b * F * heads else B
            if (b_eff, k_eff) != (b * F * heads, n * l):
                bad.append(f"{l} tokens per image: bias [{b_eff}, {n * l}, {k_eff}] vs required [{b * F * heads}, {n * l}, {n * l}]")
        if bad:
            raise ValueError("attention_mask " + str(tuple(attention_mask.shape)) + " is inconsistent with the attention shapes of this UNet, as it "
                             "is in the reference (its xformers call raises): " + "; ".join(bad))
        raise NotImplementedError("a key bias for a UNet that attends at a single resolution is not implemented in the attention kernels")

    def _forward_impl(self, sample, timestep, encoder_hidden_states, *, packed, timestep_cond, attention_mask, cross_attention_kwargs,
                      added_cond_kwargs, down_block_additional_residuals, mid_block_additional_residual, return_dict, camera, num_views,
                      i2v_cond_time_zero):
        assert sample.shape[0] % num_views == 0, "[UNet] input batch size must be dividable by num_views!"
        if timestep_cond is not None:
            # the reference adds time_embedding.cond_proj(timestep_cond) (:726-730); the SD1.5 UNet has no cond_proj (time_cond_proj_dim = None)
            raise ValueError("timestep_cond needs a time_embedding.cond_proj, which this UNet (time_cond_proj_dim = None) does not have")
        if attention_mask is not None:
            self._reject_attention_mask(attention_mask, sample, num_views)
        if cross_attention_kwargs:
            # diffusers hands `scale` to the processors as the LoRA scale; without LoRA layers (this model has none) it changes nothing
            extra = set(cross_attention_kwargs) - {"scale"}
            if extra:
                raise NotImplementedError(f"cross_attention_kwargs {sorted(extra)} are not supported (only the LoRA `scale`, a no-op here)")
        residuals = down_block_additional_residuals is not None or mid_block_additional_residual is not None
        if residuals and (torch.is_grad_enabled() and getattr(self, "_training_enabled", False)):
            raise NotImplementedError("ControlNet residuals are an inference input (:787-796, 816-817); the training path does not differentiate them")
        if residuals and self.parallel is not None and self.parallel.world > 1:
            raise NotImplementedError("ControlNet residuals with a sharded (shard_unet) model are not supported")
        cfg, ops = self.config, self.ops
        V, _, F, H, W = sample.shape
        n = self.num_views or num_views
        if V % n != 0:
            raise AssertionError("[UNet] input batch size must be dividable by the processors' num_views!")
        nlev = len(cfg.block_out_channels)
        if min(H, W) < (1 << (nlev - 1)):
            raise ValueError(f"latent size {(H, W)} is smaller than the {1 << (nlev - 1)}x total down-sampling")
        if F > cfg.motion_max_seq_length:
            raise ValueError(f"num_frames {F} exceeds motion_max_seq_length {cfg.motion_max_seq_length}")
        P = packed if packed is not None else (self._packed if self._packed is not None else self._pack())
        dev, adt = sample.device, ops.act_dtype
        img_embeds = None if added_cond_kwargs is None else added_cond_kwargs.get("image_embeds")
        if torch.is_tensor(timestep) and timestep.numel() not in (1, V):
            raise ValueError(f"timestep must be a scalar or have one entry per video, got {tuple(timestep.shape)}")
        # multi-GPU: cut this rank's (b, view, frame) shard out of the full call (animate3d_amd.parallel)
        par = self.parallel if (self.parallel is not None and self.parallel.world > 1) else None
        V_full, n_full, F_full, f0 = V, n, F, 0
        if par is not None:
            par.configure(V // n, n, F)
            idx = par.local_videos_on(V, n, dev)
            f0, F = par.frame_range(F)
            sample = sample.index_select(0, idx)[:, :, f0:f0 + F]
            encoder_hidden_states = encoder_hidden_states.to(dev).index_select(0, idx)
            camera = None if camera is None else camera.to(dev).index_select(0, idx)
            img_embeds = None if img_embeds is None else img_embeds.to(dev).index_select(0, idx)
            if torch.is_tensor(timestep) and timestep.numel() == V:
                timestep = timestep.to(dev).reshape(-1).index_select(0, idx)
            V, n = idx.numel(), n // par.view_shards
        self._frames = (F_full, f0)          # all frames of the call / first frame of this rank (temporal encoding, frame exchange)
        B2 = V * F

        # 1. time / camera embedding (unet_motion_mv_model.py:706-752)
        if not torch.is_tensor(timestep):
            t = torch.full((V,), float(timestep), dtype=torch.float32, device=dev)
        else:
            t = timestep.to(device=dev, dtype=torch.float32).reshape(-1).expand(V).contiguous()

        def time_mlp(tt):
            e = ops.gemm(ops.timestep_embed(tt, cfg.block_out_channels[0]), P.time[0], P.time[1])
            return ops.gemm(ops.silu(e), P.time[2], P.time[3])

        emb = time_mlp(t)
        cond_emb = time_mlp(torch.zeros_like(t)) if i2v_cond_time_zero else None
        if camera is not None:
            assert camera.shape[0] == V
            if P.cam is None:
                raise ValueError("camera passed but the model has no camera_embedding")
            cam = torch.zeros((V, 64), dtype=adt, device=dev)
            cam[:, : camera.shape[1]] = camera.to(device=dev, dtype=adt)
            c1 = ops.silu(ops.gemm(cam, P.cam[0], P.cam[1]))
            emb = ops.gemm(c1, P.cam[2], P.cam[3], residual=emb)
            if cond_emb is not None:
                cond_emb = ops.gemm(c1, P.cam[2], P.cam[3], residual=cond_emb)
        if cond_emb is None:
            semb, rb_rows = ops.silu(emb), F                 # one embedding per video, F images share it
        else:                                               # frame 0 of every video gets the t=0 embedding (:748-752)
            per_img = emb[:, None, :].repeat(1, F, 1)
            if f0 == 0:
                per_img[:, 0] = cond_emb
            semb, rb_rows = ops.silu(per_img.reshape(B2, -1).contiguous()), 1

        # 2. conditioning tokens, once per video (reference repeats them per frame, :754-764)
        T = encoder_hidden_states.shape[1]
        text_rows = encoder_hidden_states.to(device=dev, dtype=adt).reshape(V * T, -1).contiguous()
        ip_rows = []
        if self.encoder_hid_proj is not None and cfg.encoder_hid_dim_type == "ip_image_proj":
            if img_embeds is None:
                raise ValueError(f"{self.__class__} has the config param `encoder_hid_dim_type` set to 'ip_image_proj' which requires the "
                                 "keyword argument `image_embeds` to be passed in  `added_conditions`")
            img = img_embeds.to(device=dev, dtype=adt).reshape(V, -1).contiguous()
            pr = ops.gemm(img, P.ip[0], P.ip[1]).reshape(V * cfg.ip_num_tokens, cfg.cross_attention_dim)
            ip_rows.append(ops.layer_norm(pr, P.ip[2], P.ip[3], P.ip[4]))

        # (inference: the conditioning-only projections of all layers in three GEMMs, _pack_conditioning)
        self._cond = None
        if self.stack_conditioning and self._active_ops is None and getattr(P, "cond", None) is not None and P is self._packed:
            self._cond = self._project_conditioning(P.cond, semb, text_rows, ip_rows)

        # 3. conv_in over im2col patches; from here on x is [(V F) h w, C] rows
        x = ops.gemm(ops.im2col_in(sample), P.conv_in[0], P.conv_in[1])
        h_, w_ = H, W
        skips = [x]
        sizes = [(H, W)]                  # feature-map size per level: the forced upsample sizes of :690-698, 831-837
        ckpt = bool(getattr(self, "_gradient_checkpointing", False)) and torch.is_grad_enabled() and self._active_ops is not None

        def layer(x_in, rp, tp, mp, hh, ww):
            """ResNet (+ Transformer2D) + motion module of one layer; under gradient checkpointing only its input and output are kept."""
            aops = self._active_ops
            def run(xx):
                prev, self._active_ops = self._active_ops, aops      # the recomputation runs inside backward(), after forward() has returned
                try:
                    xx = self._resnet(xx, B2, hh, ww, rp, semb, rb_rows)
                    if tp is not None:
                        xx = self._t2d(xx, V, n, F, hh, ww, tp, text_rows, ip_rows, T)
                    return self._motion(xx, V, n, F, hh, ww, mp)
                finally:
                    self._active_ops = prev
            if ckpt:
                from torch.utils.checkpoint import checkpoint
                return checkpoint(run, x_in, use_reentrant=False, preserve_rng_state=False)
            return run(x_in)

        for pk in P.down:
            for j, rp in enumerate(pk.resnets):
                x = layer(x, rp, pk.t2d[j] if pk.t2d is not None else None, pk.motion[j], h_, w_)
                skips.append(x)
            if pk.down is not None:
                x, h_, w_ = ops.conv3x3(x, B2, h_, w_, pk.down[0], pk.down[1], stride=2)
                skips.append(x)
                sizes.append((h_, w_))
        def plus(x_rows, res, hh, ww):
            """rows + a ControlNet residual given as the reference gives it, [(V F), C, hh, ww] (:787-796, 816-817)."""
            C = x_rows.shape[1]
            if tuple(res.shape) != (B2, C, hh, ww):
                raise ValueError(f"additional residual of shape {tuple(res.shape)}, expected {(B2, C, hh, ww)}")
            r = ops.empty(B2 * hh * ww, C)                                  # a fresh buffer: the sum is formed in it (never in the caller's tensor)
            r.copy_(res.to(device=dev).permute(0, 2, 3, 1).reshape(B2 * hh * ww, C))      # layout change to token rows
            return ops.axpby_(x_rows, r, 1.0, 1.0)

        if down_block_additional_residuals is not None:
            if len(down_block_additional_residuals) != len(skips):
                raise ValueError(f"{len(down_block_additional_residuals)} down-block residuals for {len(skips)} skip connections")
            geo = [sizes[0]]
            for bi, pkd in enumerate(P.down):
                geo += [sizes[bi]] * len(pkd.resnets) + ([sizes[bi + 1]] if pkd.down is not None else [])
            skips = [plus(s_, r_, *g_) for s_, r_, g_ in zip(skips, down_block_additional_residuals, geo)]
        pk = P.mid
        x = layer(x, pk.resnets[0], pk.t2d[0], pk.motion[0], h_, w_)
        x = self._resnet(x, B2, h_, w_, pk.resnets[1], semb, rb_rows)
        if mid_block_additional_residual is not None:
            x = plus(x, mid_block_additional_residual, h_, w_)
        lvl = len(sizes) - 1
        for bi, (blk, pk) in enumerate(zip(self.up_blocks, P.up)):
            for j, rp in enumerate(pk.resnets):
                skip = skips.pop()
                if getattr(self, "_freeu_factors", None) is not None:
                    x, skip = self._freeu(bi, x, skip, B2, h_, w_)
                # inference on an op set with the two-source kernels: no concatenation pass (the pair goes to _resnet); training forwards
                # (autograd op set, checkpointed layers take one tensor) keep the concat
                pair = self._active_ops is None and hasattr(ops, "gemm2") and hasattr(ops, "group_norm2")
                x = layer((x, skip) if pair else ops.concat(x, skip), rp, pk.t2d[j] if pk.t2d is not None else None, pk.motion[j], h_, w_)
            if pk.up is not None:
                # the reference passes the next skip's size whenever a latent side is not a multiple of 2^(levels-1);
                # when it is, that size is the plain 2x, so always naming it is the same arithmetic
                lvl -= 1
                x, h_, w_ = ops.conv3x3(x, B2, h_, w_, pk.up[0], pk.up[1], up2x=True, up_size=sizes[lvl])
        x = ops.group_norm(x, B2, h_ * w_, P.norm_out[0], P.norm_out[1], cfg.norm_num_groups, cfg.norm_eps, True)
        x, _, _ = ops.conv3x3(x, B2, h_, w_, P.conv_out[0], P.conv_out[1])
        out_dtype = sample.dtype if sample.dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32
        out = ops.unpack_out(x, V, cfg.out_channels, F, H, W, out_dtype)
        if par is not None:
            out = par.all_gather_output(out, V_full, n_full, F_full)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)
