"""ctypes binding of libanimate3d_hip.so (C-ABI in include/animate3d_hip.h) + the op set the
UNet host code is written against.

torch is used here only for device memory (``torch.empty``), the current HIP stream handle and
pointer extraction; every arithmetic op goes through the C-ABI.  There is NO fallback: if the
library is missing, fails to load, or a tensor is not a CUDA tensor of the op set's 16-bit storage type, these calls raise.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libanimate3d_hip.so")
_lib = None

c_i64, c_int, c_f32, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class _RowMapC(ctypes.Structure):
    _fields_ = [("gdiv", c_i64), ("ga", c_i64), ("gb", c_i64), ("seg_len", c_i64), ("seg_stride", c_i64), ("ld", c_i64)]


@dataclass(frozen=True)
class RowMap:
    """row(g, s) = (g // gdiv) * ga + (g % gdiv) * gb + (s // seg_len) * seg_stride + s % seg_len
    (include/animate3d_hip.h: a3d_rowmap).  ``ld`` comes from the tensor it is applied to."""
    gdiv: int
    ga: int
    gb: int
    seg_len: int
    seg_stride: int

    def c(self, ld: int) -> _RowMapC:
        return _RowMapC(self.gdiv, self.ga, self.gb, self.seg_len, self.seg_stride, ld)


def rowmap_rows(m: RowMap, groups: int, step: int, length: int, device="cpu") -> torch.Tensor:
    """Rows ``m`` addresses for groups 0, step, 2 step, ... < groups and positions 0 .. length-1, in (group, position) order."""
    g = torch.arange(0, groups, step, device=device).view(-1, 1)
    s = torch.arange(length, device=device).view(1, -1)
    return ((g // m.gdiv) * m.ga + (g % m.gdiv) * m.gb + (s // m.seg_len) * m.seg_stride + s % m.seg_len).reshape(-1)


def rowmap_covers(m: RowMap, groups: int, step: int, length: int, rows: int, device="cpu") -> bool:
    """True when those rows are ALL of 0 .. rows-1, each exactly once: an attention backward whose key map covers K / V writes every row of
    dK / dV, so they need no zero fill (the first-frame maps, which read frame 0 only, do not cover)."""
    if (groups + step - 1) // step * length != rows:
        return False
    idx = rowmap_rows(m, groups, step, length, device)
    return bool(int(idx.min()) >= 0 and int(idx.max()) < rows and torch.unique(idx).numel() == rows)


# symbol -> (restype, argtypes); mirrors include/animate3d_hip.h line by line
_SIGNATURES = {
    "a3d_version": (ctypes.c_char_p, []),
    "a3d_gemm_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_int]),
    "a3d_gemm_ws_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_f32, c_f32, c_int,
                                 c_vp, c_i64, ctypes.POINTER(c_i64)]),
    "a3d_gemm2_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int]),
    "a3d_gemm_geglu_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int]),
    "a3d_conv3x3_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "a3d_conv3x3_ws_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_vp, c_i64, ctypes.POINTER(c_i64)]),
    "a3d_flash_attn_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC),
                                    c_int, c_int, c_int, c_i64, c_i64, c_f32, c_f32, c_int]),
    "a3d_flash_attn_counted_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC),
                                            c_int, c_int, c_int, c_i64, c_i64, c_f32, c_f32, c_int, c_vp]),
    "a3d_flash_attn2_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC),
                                     ctypes.POINTER(_RowMapC), c_int, c_int, c_int, c_i64, c_i64, c_i64, c_f32, c_f32, c_f32, c_int]),
    "a3d_temporal_attn_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_i64, c_int, c_int, c_f32]),
    "a3d_temporal_attn_sharded_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_i64, c_int, c_int, c_f32,
                                               c_int, c_int, c_int, c_i64]),
    "a3d_group_norm_ws_floats": (c_i64, [c_int, c_i64, c_int]),
    "a3d_group_norm_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_int, c_f32, c_int]),
    "a3d_group_norm2_bf16": (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_f32, c_int]),
    "a3d_group_norm_sums_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_int]),
    "a3d_group_norm_apply_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_int, c_int]),
    "a3d_layer_norm_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_f32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64]),
    "a3d_geglu_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64]),
    "a3d_silu_bf16": (c_int, [c_vp, c_vp, c_vp, c_i64]),
    "a3d_activation_bf16": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int]),
    "a3d_concat_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64]),
    "a3d_timestep_embed_bf16": (c_int, [c_vp, c_vp, c_vp, c_int, c_int]),
    "a3d_im2col_in": (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int]),
    "a3d_unpack_out": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int]),
    "a3d_gemm_f32out_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_f32]),
    "a3d_softmax_rows_f32_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64]),
    "a3d_channel_mix_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_i64, c_f32]),
    "a3d_cfg_ddim_step_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_i64, c_f32, c_f32, c_f32]),
    # training path (include/animate3d_hip.h, "Training path" section)
    "a3d_flash_attn_bwd_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC),
                                        ctypes.POINTER(_RowMapC), c_int, c_int, c_int, c_i64, c_i64, c_int, c_f32, c_f32, c_int]),
    "a3d_flash_attn_lse_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC),
                                        c_int, c_int, c_int, c_i64, c_i64, c_f32, c_f32, c_int, c_vp]),
    "a3d_attn_delta_bf16": (c_int, [c_vp, c_vp, c_vp, ctypes.POINTER(_RowMapC), ctypes.POINTER(_RowMapC), c_vp, c_int, c_int, c_int, c_i64]),
    "a3d_temporal_attn_bwd_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_i64, c_int, c_int, c_f32]),
    "a3d_layer_norm_bwd_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_f32, c_int]),
    "a3d_group_norm_bwd_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_int, c_int]),
    "a3d_geglu_bwd_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64]),
    "a3d_transpose_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64]),
    "a3d_wgrad_ws_floats": (c_i64, [c_i64, c_i64, c_i64]),
    "a3d_wgrad_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_f32, c_int]),
    "a3d_colsum_bf16": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_f32, c_int]),
    "a3d_axpby_bf16": (c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_f32]),
    "a3d_zero_insert2x_bf16": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int]),
    "a3d_upsample2x_bwd_bf16": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int]),
    "a3d_sqnorm_f32": (c_int, [c_vp, c_vp, c_i64, c_vp, c_int]),
    "a3d_clip_ctrl_f32": (c_int, [c_vp, c_vp, c_f32, c_f32, c_vp]),
    "a3d_adamw_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp]),
}
# fp16-storage twins (include/animate3d_hip.h, last section): same signatures
for _name in list(_SIGNATURES):
    if _name.endswith("_bf16"):
        _SIGNATURES[_name[:-5] + "_f16"] = _SIGNATURES[_name]
    elif _name in ("a3d_im2col_in", "a3d_unpack_out"):
        _SIGNATURES[_name + "_f16"] = _SIGNATURES[_name]
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
A3D_EUNSUPPORTED = -2          # include/animate3d_hip.h


def lib_path() -> str:
    return _LIB_PATH


def load_library():
    """dlopen the in-tree library and type every entry point.  Raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `python -m animate3d_amd.build` (or __graft_entry__.build()). "
                               "There is no CPU fallback for the MI355X kernels.")
        lib = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "A3D_EINVAL (shape/alignment precondition)", -2: "A3D_EUNSUPPORTED"}.get(rc, f"hipError {rc}")
        raise RuntimeError(f"{what} failed: {kind}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def on_model_device(method):
    """Decorator for the entry points of a module that owns a ``HipOps``: the C-ABI launches on the CURRENT device's stream
    (``torch.cuda.current_stream``) and allocates there, so a model living on cuda:1 must run with cuda:1 current.  One
    process per GPU is the deployment model; this makes a second device in one process work as well."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        dev = self.device
        if dev.type == "cuda" and torch.cuda.current_device() != (dev.index if dev.index is not None else torch.cuda.current_device()):
            with torch.cuda.device(dev):
                return method(self, *args, **kwargs)
        return method(self, *args, **kwargs)
    return wrapper


class _Entry:
    """Entry points of one storage type: attribute ``a3d_gemm_bf16`` resolves to ``a3d_gemm_f16`` for an fp16 op set."""

    def __init__(self, lib, f16: bool):
        self._lib, self._f16 = lib, f16

    def __getattr__(self, name):
        if self._f16:
            if name.endswith("_bf16"):
                name = name[:-5] + "_f16"
            elif name in ("a3d_im2col_in", "a3d_unpack_out"):
                name = name + "_f16"
        return getattr(self._lib, name)


class HipOps:
    """The op set of the denoise step, executed by the gfx950 kernels.  All activations are 2-D ``[rows, C]`` CUDA tensors
    (NHWC images flattened to rows) in the op set's storage type: bf16 (default) or IEEE fp16 (``act_dtype=torch.float16``:
    what a model cast with ``.half()`` gets, as the reference's 4D-SDS caller does, animatemv_guidance.py:339-346).  Both use
    fp32 accumulation, statistics and softmax; an fp16 model keeps 11 significant bits where bf16 keeps 8."""

    has_attn_lse = True      # flash_attn(with_lse=True) / flash_attn_bwd(o=..., lse=...): see autograd_ops._FlashAttn

    def __init__(self, device: Optional[torch.device] = None, act_dtype: torch.dtype = torch.bfloat16, split_k: bool = True):
        """``split_k`` = False: only kernels whose results do not depend on the launch shape (bit-for-bit batch independence, a sharded forward
        bit-equal to the unsharded one); True (default): the small-M / long-K convolutions and linears split their K sum over idle CUs —
        deterministic per launch shape, equal to the unsplit kernels to fp32 summation order."""
        if not torch.cuda.is_available():
            raise RuntimeError("HipOps needs a visible MI355X (torch.cuda.is_available() is False); no CPU fallback exists")
        if act_dtype not in (torch.bfloat16, torch.float16):
            raise ValueError(f"storage type {act_dtype} is not supported (bfloat16 or float16)")
        self.raw_lib = load_library()
        self._kv_row_cache = {}
        self._kv_cover_cache = {}      # _shared_kv_rows
        self.act_dtype = act_dtype
        self.lib = _Entry(self.raw_lib, act_dtype == torch.float16)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # compute units the persistent GEMM / conv kernels leave free for concurrently running kernels: passed with EVERY launch (the flags
        # word of a3d_gemm / a3d_gemm_geglu / a3d_conv3x3) — the library keeps no state.  Set by animate3d_amd.parallel while an RCCL
        # all-gather is in flight; a HIP graph captured meanwhile replays with the value of its capture.
        self.reserved_cus = 0
        # split-K of the small-M / long-K GEMMs and convs (UNet levels 2 / 3, the whole 4D-SDS shape): on by default; results then agree
        # with the unsplit kernels to fp32 summation order only, so runs that are compared BIT FOR BIT across different row counts
        # (a sharded forward against the unsharded one: animate3d_amd.parallel turns it off) must not use it
        self.split_k = bool(split_k)
        # ... of dense GEMMs too: off.  Measured (profiles/r5_microbench_splitk.log): the mid-block convolutions (K = 11 520 ... 23 040) gain
        # 13-55 %, but the level-2 / 3 linears (K <= 5 120, 35-120 us per launch) LOSE 20-100 %: 84 MB of fp32 partials written and read
        # back plus a second launch cost more than the idle CUs return.  The entry point and its tests stay (a3d_gemm_ws_*).
        self.split_k_gemm = False
        # diagnostics (a3d_flash_attn_counted): an int32 device tensor of 3 words that every plain ``flash_attn`` call adds to while it is set —
        # [0] workgroups of the LDS-DMA staged attention kernels sent to the exact pass by the fp16 spread vote, [1] exact re-runs after an
        # overflow of the max-free pass, [2] workgroups launched.  None (default): the plain entry point.
        self.attn_counters = None
        self._ws_plan = {}           # launch shape -> split-K workspace bytes (0: the shape does not split)
        self._ws_buf = None          # ONE split-K workspace per op set, grown to the largest need (launches on one stream run in order,
                                     # so consecutive split launches may share it; a fresh 100-MB torch.empty per call was the alternative)

    def _workspace(self, need: int) -> torch.Tensor:
        if self._ws_buf is None or self._ws_buf.numel() < need:
            self._ws_buf = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
        return self._ws_buf

    def _gemm_flags(self, tile128: bool, ring: bool = False, direct: bool = False) -> int:
        r = int(self.reserved_cus)
        if not 0 <= r <= 0xFF:
            raise ValueError(f"reserved_cus = {self.reserved_cus}: the flags word of a3d_gemm / a3d_conv3x3 carries 0..255 reserved compute units")
        return r | (0x100 if tile128 else (0x200 if ring else (0x300 if direct else 0)))

    # ---- helpers
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _act(self, t: torch.Tensor, name: str):
        if t.dtype != self.act_dtype or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError(f"{name}: expected a 2-D {self.act_dtype} CUDA tensor with contiguous rows, got {t.dtype} {tuple(t.shape)} {t.device} strides {t.stride()}")
        return t

    def split_cols(self, x, *bounds):
        """The column pieces [bounds[i], bounds[i+1]) of a fused projection output as views (the differentiable op set overrides
        this: autograd_ops.AutogradOps.split_cols collects the pieces' gradients in one buffer)."""
        return tuple(x[:, a:b] for a, b in zip(bounds[:-1], bounds[1:]))

    def empty(self, rows: int, cols: int) -> torch.Tensor:
        return torch.empty((rows, cols), dtype=self.act_dtype, device=self.device)

    # ---- GEMM family
    def gemm(self, x, w, bias=None, *, residual=None, alpha: float = 1.0, beta: float = 1.0, rowbias=None, rb_div: int = 1, out=None,
             tile128: bool = False, ring: bool = False, direct: bool = False):
        """``tile128`` forces the register-staged 128 x 128-tile kernel (A3D_GEMM_TILE128), ``ring`` the LDS-DMA ring kernel where the shape
        allows it (A3D_GEMM_RING), ``direct`` the persistent kernel's LDS-free epilogue (A3D_GEMM_DIRECT): bit-identical results; parity tests
        and A/B timing."""
        x, w = self._act(x, "gemm.x"), self._act(w, "gemm.w")
        M, K = x.shape
        N = w.shape[0]
        assert w.shape[1] == K, (x.shape, w.shape)
        y = out if out is not None else self.empty(M, N)
        self._act(y, "gemm.out")
        if residual is not None:
            self._act(residual, "gemm.residual")
            assert residual.shape == (M, N)
        if rowbias is not None:
            self._act(rowbias, "gemm.rowbias")
            assert rowbias.is_contiguous() and rowbias.shape[1] == N
        if bias is not None:
            assert bias.dtype == torch.float32 and bias.numel() == N
        flags = self._gemm_flags(tile128, ring, direct)
        args = (self._stream(), _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(rowbias), rb_div,
                _p(residual), residual.stride(0) if residual is not None else 0, _p(y), y.stride(0), M, N, K, alpha, beta, flags)
        if self.split_k and self.split_k_gemm and not tile128 and M <= 32768 and K >= 1152:
            key = ("gemm", M, N, K, x.stride(0), w.stride(0), y.stride(0), rb_div if rowbias is not None else 0, flags)
            need = self._ws_plan.get(key)
            if need is None:
                q = c_i64(0)
                _check(self.lib.a3d_gemm_ws_bf16(*args, None, 0, ctypes.byref(q)), f"a3d_gemm_ws_bf16 (query) M={M} N={N} K={K}")
                need = self._ws_plan[key] = int(q.value)
            if need > 0:
                ws = self._workspace(need)
                _check(self.lib.a3d_gemm_ws_bf16(*args, _p(ws), need, None), f"a3d_gemm_ws_bf16 M={M} N={N} K={K}")
                return y
        rc = self.lib.a3d_gemm_bf16(*args)
        _check(rc, f"a3d_gemm_bf16 M={M} N={N} K={K}")
        return y

    def gemm2(self, xa, xb, w, bias=None):
        """``[xa | xb] w^T + bias`` without the concatenation (a3d_gemm2: the 1x1 shortcut conv of an up-block ResNet).  None when the
        persistent kernel does not take the shape — the caller then concatenates and calls ``gemm``."""
        xa, xb, w = self._act(xa, "gemm2.xa"), self._act(xb, "gemm2.xb"), self._act(w, "gemm2.w")
        M, K1 = xa.shape
        K = K1 + xb.shape[1]
        N = w.shape[0]
        assert xb.shape[0] == M and w.shape[1] == K
        if K1 % 64 or K % 64 or N % 8:
            return None
        y = self.empty(M, N)
        rc = self.lib.a3d_gemm2_bf16(self._stream(), _p(xa), xa.stride(0), _p(xb), xb.stride(0), K1, _p(w), w.stride(0), _p(bias), _p(y), y.stride(0),
                                     M, N, K, self._gemm_flags(False))
        if rc == A3D_EUNSUPPORTED:
            return None
        _check(rc, f"a3d_gemm2_bf16 M={M} N={N} K={K1}+{K - K1}")
        return y

    @staticmethod
    def interleave_geglu(w: torch.Tensor) -> torch.Tensor:
        """[2N, ...] = [h rows | gate rows] -> rows interleaved in blocks of 32 (layout a3d_gemm_geglu_bf16 expects)."""
        n = w.shape[0] // 2
        assert n % 32 == 0
        h, g = w[:n].reshape(n // 32, 32, *w.shape[1:]), w[n:].reshape(n // 32, 32, *w.shape[1:])
        return torch.stack([h, g], dim=1).reshape(w.shape).contiguous()

    def gemm_geglu(self, x, w_il, bias_il, *, tile128: bool = False):
        """GEGLU(x) = (x Wh^T + bh) * gelu(x Wg^T + bg) with interleaved weights (see interleave_geglu)."""
        x, w_il = self._act(x, "geglu.x"), self._act(w_il, "geglu.w")
        M, K = x.shape
        N2 = w_il.shape[0]
        y = self.empty(M, N2 // 2)
        rc = self.lib.a3d_gemm_geglu_bf16(self._stream(), _p(x), x.stride(0), _p(w_il), w_il.stride(0), _p(bias_il), _p(y), y.stride(0), M, N2, K,
                                          self._gemm_flags(tile128))
        _check(rc, f"a3d_gemm_geglu_bf16 M={M} N2={N2} K={K}")
        return y

    def conv3x3(self, x, B: int, H: int, W: int, w, bias, *, stride: int = 1, up2x: bool = False, rowbias=None, rb_div: int = 1, residual=None,
                up_size=None, tile128: bool = False):
        """x [B*H*W, Cin] -> (y [B*Ho*Wo, Cout], Ho, Wo); w packed [Cout, 9*Cin] (ky, kx, ci).  ``up_size`` = (Ho, Wo) forces
        the size of the nearest upsampling in front of the conv (2H or 2H-1, likewise W: unet_motion_mv_model.py:831-837)."""
        x, w = self._act(x, "conv.x"), self._act(w, "conv.w")
        Cin = x.shape[1]
        Cout = w.shape[0]
        assert x.is_contiguous() and w.is_contiguous() and x.shape[0] == B * H * W and w.shape[1] == 9 * Cin
        He, We = (2 * H, 2 * W) if up2x else (H, W)
        up_code = 1 if up2x else 0
        if up_size is not None:
            if not up2x or up_size[0] not in (2 * H, 2 * H - 1) or up_size[1] not in (2 * W, 2 * W - 1):
                raise ValueError(f"forced upsample size {tuple(up_size)} is not reachable from {(H, W)} (2x or 2x - 1 per axis)")
            He, We = int(up_size[0]), int(up_size[1])
            up_code |= (2 if He == 2 * H - 1 else 0) | (4 if We == 2 * W - 1 else 0)
        Ho, Wo = (He - 1) // stride + 1, (We - 1) // stride + 1
        y = self.empty(B * Ho * Wo, Cout)
        if residual is not None:
            assert residual.is_contiguous() and residual.shape == y.shape
        if rowbias is not None:
            assert rowbias.is_contiguous() and rowbias.shape[1] == Cout
        flags = self._gemm_flags(tile128)
        args = (self._stream(), _p(x), _p(w), _p(bias), _p(rowbias), rb_div, _p(residual), _p(y), B, H, W, Cin, Cout, stride, up_code, flags)
        if self.split_k and not tile128 and B * Ho * Wo <= 32768:
            key = ("conv", B, H, W, Cin, Cout, stride, up_code, rb_div if rowbias is not None else 0, flags)
            need = self._ws_plan.get(key)
            if need is None:
                q = c_i64(0)
                _check(self.lib.a3d_conv3x3_ws_bf16(*args, None, 0, ctypes.byref(q)), f"a3d_conv3x3_ws_bf16 (query) B={B} H={H} W={W} Cin={Cin} Cout={Cout}")
                need = self._ws_plan[key] = int(q.value)
            if need > 0:
                ws = self._workspace(need)
                _check(self.lib.a3d_conv3x3_ws_bf16(*args, _p(ws), need, None), f"a3d_conv3x3_ws_bf16 B={B} H={H} W={W} Cin={Cin} Cout={Cout}")
                return y, Ho, Wo
        rc = self.lib.a3d_conv3x3_bf16(*args)
        _check(rc, f"a3d_conv3x3_bf16 B={B} H={H} W={W} Cin={Cin} Cout={Cout} stride={stride} up={up2x}")
        return y, Ho, Wo

    # ---- attention
    def flash_attn(self, q, k, v, qmap: RowMap, kmap: RowMap, groups: int, heads: int, q_len: int, kv_len: int, *,
                   out=None, out_scale: float = 1.0, accumulate: bool = False, causal: bool = False, with_lse: bool = False,
                   accumulation_target: bool = False, exact: bool = False, plain: bool = False):
        """``exact`` (A3D_ATTN_EXACT): the LDS-DMA staged kernels skip their max-free first pass; ``plain`` (A3D_ATTN_PLAIN): the generic
        kernel of the short / ragged shapes — per-call choices for the parity tests and A/B timing, results agree to rounding.
        ``accumulation_target`` is a hint for the autograd op set (autograd_ops.AutogradOps.flash_attn), ignored here.  ``with_lse``: returns (o, lse2) with lse2 [groups, heads, q_len] fp32 = log2 of every query's softmax denominator (training:
        ``flash_attn_bwd(stats=...)`` then skips its statistics pass)."""
        q, k, v = self._act(q, "attn.q"), self._act(k, "attn.k"), self._act(v, "attn.v")
        C = q.shape[1]
        D = C // heads
        assert k.stride(0) == v.stride(0)
        o = out if out is not None else self.empty(q.shape[0], C)
        qm, km, om = qmap.c(q.stride(0)), kmap.c(k.stride(0)), qmap.c(o.stride(0))
        flags = (1 if accumulate else 0) | (2 if causal else 0) | (4 if exact else 0) | (8 if plain else 0)
        if with_lse:
            lse = torch.empty((groups, heads, q_len), dtype=torch.float32, device=self.device)
            rc = self.lib.a3d_flash_attn_lse_bf16(self._stream(), _p(q), _p(k), _p(v), _p(o), ctypes.byref(qm), ctypes.byref(km), ctypes.byref(om),
                                                  groups, heads, D, q_len, kv_len, float(D) ** -0.5, out_scale,
                                                  flags, _p(lse))
            _check(rc, f"a3d_flash_attn_lse_bf16 groups={groups} heads={heads} D={D} q_len={q_len} kv_len={kv_len}")
            return o, lse
        if self.attn_counters is not None:
            cnt = self.attn_counters
            assert cnt.dtype == torch.int32 and cnt.is_cuda and cnt.numel() >= 3 and cnt.is_contiguous()
            rc = self.lib.a3d_flash_attn_counted_bf16(self._stream(), _p(q), _p(k), _p(v), _p(o), ctypes.byref(qm), ctypes.byref(km), ctypes.byref(om),
                                                      groups, heads, D, q_len, kv_len, float(D) ** -0.5, out_scale, flags, _p(cnt))
            _check(rc, f"a3d_flash_attn_counted_bf16 groups={groups} heads={heads} D={D} q_len={q_len} kv_len={kv_len}")
            return o
        rc = self.lib.a3d_flash_attn_bf16(self._stream(), _p(q), _p(k), _p(v), _p(o), ctypes.byref(qm), ctypes.byref(km), ctypes.byref(om),
                                          groups, heads, D, q_len, kv_len, float(D) ** -0.5, out_scale, flags)
        _check(rc, f"a3d_flash_attn_bf16 groups={groups} heads={heads} D={D} q_len={q_len} kv_len={kv_len}")
        return o

    def flash_attn2(self, q, k, v, k2, v2, qmap: RowMap, kmap: RowMap, kmap2: RowMap, groups: int, heads: int, q_len: int, kv_len: int,
                    kv_len2: int, *, out_scale2: float = 1.0):
        """attn(q, k, v) + out_scale2 * attn(q, k2, v2) in one launch (each key set with its own softmax); None when the head
        dimension has no fused kernel (the caller then issues two ``flash_attn`` calls)."""
        D = q.shape[1] // heads
        if D not in (40, 80):
            return None
        q, k, v, k2, v2 = self._act(q, "attn.q"), self._act(k, "attn.k"), self._act(v, "attn.v"), self._act(k2, "attn.k2"), self._act(v2, "attn.v2")
        assert k.stride(0) == v.stride(0) and k2.stride(0) == v2.stride(0)
        o = self.empty(q.shape[0], q.shape[1])
        qm, km, km2, om = qmap.c(q.stride(0)), kmap.c(k.stride(0)), kmap2.c(k2.stride(0)), qmap.c(o.stride(0))
        rc = self.lib.a3d_flash_attn2_bf16(self._stream(), _p(q), _p(k), _p(v), _p(k2), _p(v2), _p(o), ctypes.byref(qm), ctypes.byref(km),
                                           ctypes.byref(km2), ctypes.byref(om), groups, heads, D, q_len, kv_len, kv_len2, float(D) ** -0.5,
                                           1.0, out_scale2, 0)
        _check(rc, f"a3d_flash_attn2_bf16 groups={groups} heads={heads} D={D} q_len={q_len} kv_len={kv_len} kv_len2={kv_len2}")
        return o

    def temporal_attn(self, q, k, v, videos: int, frames: int, L: int, heads: int, *, q_f0: int = 0, q_frames: Optional[int] = None,
                      out=None):
        """Unsharded: q / k / v rows ((v F + f) L + l).  Frame-sharded (``q_frames`` < ``frames``): q holds this rank's frames
        [q_f0, q_f0 + q_frames) only; k / v are the all-gathered tensors ``[frames / q_frames blocks, videos * q_frames * L, C]``."""
        q, k, v = self._act(q, "tattn.q"), self._act(k, "tattn.k"), self._act(v, "tattn.v")
        C = q.shape[1]
        D = C // heads
        assert k.stride(0) == v.stride(0)
        o = self._act(out, "tattn.out") if out is not None else self.empty(q.shape[0], C)
        assert o.shape == (q.shape[0], C)
        if q_frames is None or q_frames == frames:
            assert q.stride(0) == k.stride(0)
            rc = self.lib.a3d_temporal_attn_bf16(self._stream(), _p(q), _p(k), _p(v), q.stride(0), _p(o), o.stride(0),
                                                 videos, frames, L, heads, D, float(D) ** -0.5)
        else:
            assert q.shape[0] == videos * q_frames * L and k.shape[0] == videos * frames * L
            rc = self.lib.a3d_temporal_attn_sharded_bf16(self._stream(), _p(q), q.stride(0), _p(k), _p(v), k.stride(0), _p(o), o.stride(0),
                                                         videos, frames, L, heads, D, float(D) ** -0.5, q_f0, q_frames, q_frames,
                                                         videos * q_frames * L)
        _check(rc, f"a3d_temporal_attn_bf16 videos={videos} frames={frames} L={L} D={D} q_f0={q_f0} q_frames={q_frames}")
        return o

    # ---- normalisation
    def group_norm(self, x, B: int, rows: int, gamma, beta, groups: int, eps: float, silu: bool):
        x = self._act(x, "gn.x")
        assert x.is_contiguous() and x.shape[0] == B * rows
        C = x.shape[1]
        y = self.empty(x.shape[0], C)
        ws = torch.empty(int(self.lib.a3d_group_norm_ws_floats(B, rows, groups)), dtype=torch.float32, device=self.device)
        rc = self.lib.a3d_group_norm_bf16(self._stream(), _p(x), _p(y), _p(gamma), _p(beta), _p(ws), B, rows, C, groups, eps, 1 if silu else 0)
        _check(rc, f"a3d_group_norm_bf16 B={B} rows={rows} C={C}")
        return y

    def group_norm2(self, xa, xb, B: int, rows: int, gamma, beta, groups: int, eps: float, silu: bool):
        """GroupNorm (+ SiLU) of ``cat([xa, xb], 1)`` read from its two parts (a3d_group_norm2)."""
        xa, xb = self._act(xa, "gn2.xa"), self._act(xb, "gn2.xb")
        assert xa.is_contiguous() and xb.is_contiguous() and xa.shape[0] == B * rows == xb.shape[0]
        Ca, Cb = xa.shape[1], xb.shape[1]
        y = self.empty(xa.shape[0], Ca + Cb)
        ws = torch.empty(int(self.lib.a3d_group_norm_ws_floats(B, rows, groups)), dtype=torch.float32, device=self.device)
        rc = self.lib.a3d_group_norm2_bf16(self._stream(), _p(xa), Ca, _p(xb), Cb, _p(y), _p(gamma), _p(beta), _p(ws), B, rows, groups, eps, 1 if silu else 0)
        _check(rc, f"a3d_group_norm2_bf16 B={B} rows={rows} C={Ca}+{Cb}")
        return y

    def group_norm_sums(self, x, B: int, rows: int, groups: int) -> torch.Tensor:
        """fp64 [B, groups, 2] = (sum, sum of squares) of this rank's rows (first half of a GroupNorm spread over ranks)."""
        x = self._act(x, "gn.x")
        assert x.is_contiguous() and x.shape[0] == B * rows
        ws = torch.empty(int(self.lib.a3d_group_norm_ws_floats(B, rows, groups)), dtype=torch.float32, device=self.device)
        sums = torch.empty((B, groups, 2), dtype=torch.float64, device=self.device)
        _check(self.lib.a3d_group_norm_sums_bf16(self._stream(), _p(x), _p(ws), _p(sums), B, rows, x.shape[1], groups),
               f"a3d_group_norm_sums_bf16 B={B} rows={rows} C={x.shape[1]}")
        return sums

    def group_norm_apply(self, x, B: int, rows: int, gamma, beta, groups: int, stats: torch.Tensor, silu: bool):
        """Second half: ``stats`` fp32 [B, groups, 2] = (mean, rstd)."""
        x = self._act(x, "gn.x")
        assert x.is_contiguous() and x.shape[0] == B * rows
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.shape == (B, groups, 2)
        y = self.empty(x.shape[0], x.shape[1])
        _check(self.lib.a3d_group_norm_apply_bf16(self._stream(), _p(x), _p(y), _p(gamma), _p(beta), _p(stats), B, rows, x.shape[1], groups,
                                                  1 if silu else 0), f"a3d_group_norm_apply_bf16 B={B} rows={rows} C={x.shape[1]}")
        return y

    def layer_norm(self, x, gamma, beta, eps: float, pe1=None, pe1_div: int = 1, pe2=None, pe2_div: int = 1, two: bool = False):
        """y1 = LN(x) + pe1[(m // pe1_div) % len(pe1)]; with two=True also y2 with pe2."""
        x = self._act(x, "ln.x")
        assert x.is_contiguous()
        M, C = x.shape
        y1 = self.empty(M, C)
        y2 = self.empty(M, C) if two else None
        for pe in (pe1, pe2):
            if pe is not None:
                self._act(pe, "ln.pe")
                assert pe.is_contiguous() and pe.shape[1] == C
        rc = self.lib.a3d_layer_norm_bf16(self._stream(), _p(x), _p(y1), _p(y2), _p(gamma), _p(beta), M, C, eps,
                                          _p(pe1), pe1_div, pe1.shape[0] if pe1 is not None else 1,
                                          _p(pe2), pe2_div, pe2.shape[0] if pe2 is not None else 1)
        _check(rc, f"a3d_layer_norm_bf16 M={M} C={C}")
        return (y1, y2) if two else y1

    # ---- elementwise / layout
    def geglu(self, x):
        x = self._act(x, "geglu.x")
        M, N2 = x.shape
        y = self.empty(M, N2 // 2)
        _check(self.lib.a3d_geglu_bf16(self._stream(), _p(x), x.stride(0), _p(y), y.stride(0), M, N2 // 2), "a3d_geglu_bf16")
        return y

    def silu(self, x):
        x = self._act(x, "silu.x")
        assert x.is_contiguous()
        y = torch.empty_like(x)
        _check(self.lib.a3d_silu_bf16(self._stream(), _p(x), _p(y), x.numel()), "a3d_silu_bf16")
        return y

    def activation(self, x, kind: str):
        """``kind``: "silu", "quick_gelu" (x * sigmoid(1.702 x)) or "gelu" (erf)."""
        x = self._act(x, "act.x")
        assert x.is_contiguous()
        y = torch.empty_like(x)
        _check(self.lib.a3d_activation_bf16(self._stream(), _p(x), _p(y), x.numel(), {"silu": 0, "quick_gelu": 1, "gelu": 2}[kind]), f"a3d_activation {kind}")
        return y

    def concat(self, a, b):
        a, b = self._act(a, "concat.a"), self._act(b, "concat.b")
        assert a.is_contiguous() and b.is_contiguous() and a.shape[0] == b.shape[0]
        y = self.empty(a.shape[0], a.shape[1] + b.shape[1])
        _check(self.lib.a3d_concat_bf16(self._stream(), _p(a), a.shape[1], _p(b), b.shape[1], _p(y), a.shape[0]), "a3d_concat_bf16")
        return y

    def timestep_embed(self, t: torch.Tensor, dim: int):
        assert t.dtype == torch.float32 and t.is_cuda and t.dim() == 1
        y = self.empty(t.shape[0], dim)
        _check(self.lib.a3d_timestep_embed_bf16(self._stream(), _p(t), _p(y), t.shape[0], dim), "a3d_timestep_embed_bf16")
        return y

    def im2col_in(self, sample: torch.Tensor):
        """[V, C, F, H, W] (fp32/bf16/fp16) -> [(V F) H W, 64] bf16 3x3 patches."""
        assert sample.is_cuda and sample.dim() == 5 and sample.dtype in _DTYPE_CODE
        sample = sample.contiguous()
        V, C, F, H, W = sample.shape
        y = self.empty(V * F * H * W, 64)
        _check(self.lib.a3d_im2col_in(self._stream(), _p(sample), _DTYPE_CODE[sample.dtype], _p(y), V, C, F, H, W), "a3d_im2col_in")
        return y

    def unpack_out(self, x, V: int, C: int, F: int, H: int, W: int, dtype: torch.dtype):
        x = self._act(x, "unpack.x")
        assert x.is_contiguous() and x.shape == (V * F * H * W, C)
        y = torch.empty((V, C, F, H, W), dtype=dtype, device=self.device)
        _check(self.lib.a3d_unpack_out(self._stream(), _p(x), _p(y), _DTYPE_CODE[dtype], V, C, F, H, W), "a3d_unpack_out")
        return y

    def gemm_f32out(self, x, w, bias=None, alpha: float = 1.0):
        """fp32 Y = alpha * (x w^T + bias) for logits that must not be rounded to bf16 (VAE mid-block attention)."""
        x, w = self._act(x, "gemm_f32out.x"), self._act(w, "gemm_f32out.w")
        M, K = x.shape
        N = w.shape[0]
        assert x.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        _check(self.lib.a3d_gemm_f32out_bf16(self._stream(), _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(y), N, M, N, K, alpha),
               f"a3d_gemm_f32out M={M} N={N} K={K}")
        return y

    def softmax_rows(self, x, out=None):
        """fp32 logits [M, N] -> bf16 row softmax (``out``: [M, N] view with its own row stride)."""
        assert x.dtype == torch.float32 and x.is_contiguous() and x.is_cuda
        M, N = x.shape
        y = self._act(out, "softmax_rows.out") if out is not None else self.empty(M, N)
        assert y.shape == (M, N)
        _check(self.lib.a3d_softmax_rows_f32_bf16(self._stream(), _p(x), N, _p(y), y.stride(0), M, N), f"a3d_softmax_rows_f32_bf16 M={M} N={N}")
        return y

    def channel_mix(self, x, w, bias, scale: float = 1.0):
        """planar fp32 [B, Cin, H, W] -> [B, Cout, H, W]: scale * (w x) + bias per pixel (<= 8 channels)."""
        assert x.dtype == torch.float32 and x.is_contiguous() and x.is_cuda
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        w = w.to(device=x.device, dtype=torch.float32).contiguous()
        b = None if bias is None else bias.to(device=x.device, dtype=torch.float32).contiguous()
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
        _check(self.lib.a3d_channel_mix_f32(self._stream(), _p(x), _p(w), _p(b), _p(y), B, Cin, Cout, H * W, scale), "a3d_channel_mix_f32")
        return y

    # ---- training path: backward kernels (reference: torch autograd behind train.py:576-590) and the optimiser step
    def flash_attn_bwd(self, q, k, v, do, qmap: RowMap, kmap: RowMap, groups: int, heads: int, q_len: int, kv_len: int, *,
                       q_per_kv: int = 1, do_scale: float = 1.0, need_dq: bool = True, need_dkv: bool = True, o=None, lse=None,
                       dq_out=None, dk_out=None, dv_out=None):
        """Gradients of ``flash_attn`` (same maps): ``do`` = gradient of the output buffer (rows as q), scaled by ``do_scale`` (= the
        forward's out_scale).  Returns (dq [q rows, C] | None, dk, dv [k rows, C] | None); rows of dk / dv that the K/V map never
        reads (e.g. frames > 0 in the first-frame branch) are zero.  ``o`` / ``lse`` (the un-accumulated output and the log-sum-exp of
        ``flash_attn(with_lse=True)``): the statistics pass is replaced by one elementwise kernel (delta = rowsum(dO * O) per head).
        ``dq_out`` / ``dk_out`` + ``dv_out``: [rows, C] views with their own row stride (the column blocks of the gradient of a fused
        Q|K|V projection output) that the kernel writes instead of fresh tensors; they are returned."""
        q, k, v, do = self._act(q, "attn_bwd.q"), self._act(k, "attn_bwd.k"), self._act(v, "attn_bwd.v"), self._act(do, "attn_bwd.do")
        C = q.shape[1]
        D = C // heads
        assert k.stride(0) == v.stride(0) and do.shape == q.shape
        def grad_buf(out, rows, covered):
            # rows the map never reaches must still come back defined (zero): the query map reaches all of them in the model's calls, the
            # K / V map does not in the first-frame branches
            if out is not None:
                assert out.shape == (rows, C) and out.stride(1) == 1 and out.dtype == self.act_dtype and out.stride(0) % 8 == 0 and out.data_ptr() % 16 == 0
                if not covered:
                    out.zero_()
                return out
            return self.empty(rows, C) if covered else torch.zeros((rows, C), dtype=self.act_dtype, device=self.device)
        assert (dk_out is None) == (dv_out is None)
        dq = grad_buf(dq_out, q.shape[0], q.shape[0] == groups * q_len) if need_dq else None
        kv_covered = need_dkv and self._kv_map_covers(kmap, groups, q_per_kv, kv_len, k.shape[0])
        dk = grad_buf(dk_out, k.shape[0], kv_covered) if need_dkv else None
        dv = grad_buf(dv_out, k.shape[0], kv_covered) if need_dkv else None
        assert dk is None or dk.stride(0) == dv.stride(0)
        qm, km, dom = qmap.c(q.stride(0)), kmap.c(k.stride(0)), qmap.c(do.stride(0))
        dqm, dkm = qmap.c(dq.stride(0) if dq is not None else C), kmap.c(dk.stride(0) if dk is not None else C)
        flags = 0
        if lse is not None:
            assert o is not None and lse.dtype == torch.float32 and lse.is_contiguous() and lse.numel() == groups * heads * q_len
            o = self._act(o, "attn_bwd.o")
            lse2, delta = lse, torch.empty(groups * heads * q_len, dtype=torch.float32, device=self.device)
            om = qmap.c(o.stride(0))
            rc = self.lib.a3d_attn_delta_bf16(self._stream(), _p(do), _p(o), ctypes.byref(dom), ctypes.byref(om), _p(delta), groups, heads, D, q_len)
            _check(rc, f"a3d_attn_delta_bf16 groups={groups} heads={heads} D={D} q_len={q_len}")
            flags = 2
        else:
            stats = torch.empty((2, groups * heads * q_len), dtype=torch.float32, device=self.device)
            lse2, delta = stats[0], stats[1]
        def run(dq_, dk_, dv_, dkm_, share, fl):
            rc = self.lib.a3d_flash_attn_bwd_bf16(self._stream(), _p(q), _p(k), _p(v), _p(do), _p(dq_), _p(dk_), _p(dv_), _p(lse2), _p(delta),
                                                  ctypes.byref(qm), ctypes.byref(km), ctypes.byref(dom), ctypes.byref(dqm), ctypes.byref(dkm_),
                                                  groups, heads, D, q_len, kv_len, share, float(D) ** -0.5, do_scale, fl)
            _check(rc, f"a3d_flash_attn_bwd_bf16 groups={groups} heads={heads} D={D} q_len={q_len} kv_len={kv_len} q_per_kv={share}")

        # Shared keys (first-frame branches: the q_per_kv = F query groups of a video read frame 0's K / V): the kernel sums their dK | dV
        # inside one workgroup per key tile, i.e. groups / q_per_kv x key tiles x heads workgroups — 16 .. 256 at the training shapes,
        # a fraction of the chip (2.5 - 3.8 x the time of the same attention with private keys at head_dim 80 / 160).  When that grid
        # cannot fill the CUs, every query group writes its own partial dK | dV (full-width grid, no sharing) and one reduction over
        # the q_per_kv partials follows (torch.sum accumulates in fp32).
        wgs_shared = (groups // q_per_kv) * heads * ((kv_len + 127) // 128)
        if need_dkv and q_per_kv > 1 and wgs_shared < 512:
            if need_dq:
                run(dq, None, None, dkm, q_per_kv, flags)
            pk = self.empty(groups * kv_len, C)
            pv = self.empty(groups * kv_len, C)
            run(None, pk, pv, RowMap(1, kv_len, 0, kv_len, 0).c(C), 1, flags | (2 if need_dq else 0))       # (the first call left the statistics in lse2 / delta)
            idx = self._shared_kv_rows(kmap, groups, q_per_kv, kv_len)
            dk.index_copy_(0, idx, pk.view(groups // q_per_kv, q_per_kv, kv_len, C).sum(dim=1).view(-1, C))
            dv.index_copy_(0, idx, pv.view(groups // q_per_kv, q_per_kv, kv_len, C).sum(dim=1).view(-1, C))
            return dq, dk, dv
        run(dq, dk, dv, dkm, q_per_kv, flags)
        return dq, dk, dv

    def _kv_map_covers(self, kmap: RowMap, groups: int, q_per_kv: int, kv_len: int, rows: int) -> bool:
        """``rowmap_covers`` for this attention's key map, decided once per map and cached — on the CPU: the answer is a function of a few
        integers, and a device-side unique / min / max with read-backs would be a host synchronisation inside the first backward of every
        map (it would also fail under HIP-graph capture of a training step)."""
        key = (kmap.gdiv, kmap.ga, kmap.gb, kmap.seg_len, kmap.seg_stride, groups, q_per_kv, kv_len, rows)
        hit = self._kv_cover_cache.get(key)
        if hit is None:
            hit = self._kv_cover_cache[key] = rowmap_covers(kmap, groups, q_per_kv, kv_len, rows, "cpu")
        return hit

    def _shared_kv_rows(self, kmap: RowMap, groups: int, q_per_kv: int, kv_len: int) -> torch.Tensor:
        """Rows of the K / V tensor that the first query group of every sharing set reads, in (set, key) order (cached per map)."""
        key = (kmap.gdiv, kmap.ga, kmap.gb, kmap.seg_len, kmap.seg_stride, groups, q_per_kv, kv_len)
        cache = self._kv_row_cache
        if key not in cache:
            cache[key] = rowmap_rows(kmap, groups, q_per_kv, kv_len, self.device)
        return cache[key]

    def temporal_attn_bwd(self, q, k, v, do, videos: int, frames: int, L: int, heads: int):
        """Returns one [rows, 3C] buffer = [dq | dk | dv]."""
        q, k, v, do = self._act(q, "tattn_bwd.q"), self._act(k, "tattn_bwd.k"), self._act(v, "tattn_bwd.v"), self._act(do, "tattn_bwd.do")
        C = q.shape[1]
        assert q.stride(0) == k.stride(0) == v.stride(0)
        d = self.empty(q.shape[0], 3 * C)
        rc = self.lib.a3d_temporal_attn_bwd_bf16(self._stream(), _p(q), _p(k), _p(v), q.stride(0), _p(do), do.stride(0),
                                                 _p(d), d.data_ptr() + 2 * C, d.data_ptr() + 4 * C, 3 * C,
                                                 videos, frames, L, heads, C // heads, float(C // heads) ** -0.5)
        _check(rc, f"a3d_temporal_attn_bwd_bf16 videos={videos} frames={frames} L={L} C={C}")
        return d

    def layer_norm_bwd(self, x, dy, gamma, eps: float, need_param: bool = True):
        x, dy = self._act(x, "ln_bwd.x"), self._act(dy, "ln_bwd.dy")
        assert x.is_contiguous() and dy.is_contiguous() and x.shape == dy.shape
        M, C = x.shape
        dx = self.empty(M, C)
        dg = torch.empty(C, dtype=torch.float32, device=self.device) if need_param else None
        db = torch.empty(C, dtype=torch.float32, device=self.device) if need_param else None
        _check(self.lib.a3d_layer_norm_bwd_bf16(self._stream(), _p(x), _p(dy), _p(gamma), _p(dx), _p(dg), _p(db), M, C, eps, 0),
               f"a3d_layer_norm_bwd_bf16 M={M} C={C}")
        return dx, dg, db

    def group_norm_stats(self, x, B: int, rows: int, groups: int, eps: float) -> torch.Tensor:
        """fp32 [B, groups, 2] = (mean, rstd) from the fp64 sums kernel (the finalisation is a [B, groups] host-side expression)."""
        sums = self.group_norm_sums(x, B, rows, groups)
        m = sums / float(rows * (x.shape[1] // groups))                         # fp64 (E[x], E[x^2]); seven small launches in all
        mean = m[..., 0]
        rstd = torch.addcmul(m[..., 1], mean, mean, value=-1.0).clamp_min_(0.0).add_(eps).rsqrt_()
        return torch.stack([mean, rstd], dim=-1).to(torch.float32)

    def group_norm_bwd(self, x, dy, B: int, rows: int, gamma, beta, groups: int, stats, silu: bool, need_param: bool = False):
        x, dy = self._act(x, "gn_bwd.x"), self._act(dy, "gn_bwd.dy")
        assert x.is_contiguous() and dy.is_contiguous() and x.shape == dy.shape and x.shape[0] == B * rows
        C = x.shape[1]
        dx = self.empty(x.shape[0], C)
        ws = torch.empty(B * C * 2, dtype=torch.float32, device=self.device)
        dg = torch.zeros(C, dtype=torch.float32, device=self.device) if need_param else None
        db = torch.zeros(C, dtype=torch.float32, device=self.device) if need_param else None
        _check(self.lib.a3d_group_norm_bwd_bf16(self._stream(), _p(x), _p(dy), _p(gamma), _p(beta), _p(stats), _p(dx), _p(ws), _p(dg), _p(db),
                                                B, rows, C, groups, 1 if silu else 0), f"a3d_group_norm_bwd_bf16 B={B} rows={rows} C={C}")
        return dx, dg, db

    def geglu_bwd(self, proj_il, dy):
        proj_il, dy = self._act(proj_il, "geglu_bwd.p"), self._act(dy, "geglu_bwd.dy")
        M, N2 = proj_il.shape
        dp = self.empty(M, N2)
        _check(self.lib.a3d_geglu_bwd_bf16(self._stream(), _p(proj_il), proj_il.stride(0), _p(dy), dy.stride(0), _p(dp), N2, M, N2 // 2),
               f"a3d_geglu_bwd_bf16 M={M} N={N2 // 2}")
        return dp

    def transpose(self, x, pad: int = 64):
        """[rows, cols] -> [cols, rows rounded up to ``pad``] (zero filled): an operand of the weight-gradient GEMM."""
        x = self._act(x, "transpose.x")
        rows, cols = x.shape
        rp = (rows + pad - 1) // pad * pad
        y = self.empty(cols, rp)
        _check(self.lib.a3d_transpose_bf16(self._stream(), _p(x), x.stride(0), _p(y), rp, rows, cols, rp), f"a3d_transpose_bf16 {rows}x{cols}")
        return y

    def wgrad(self, dy, x, alpha: float = 1.0):
        """fp32 [N, K] = alpha * dy^T x: the weight gradient of y = x W^T (split over the token axis, fp32 atomics)."""
        dy, x = self._act(dy, "wgrad.dy"), self._act(x, "wgrad.x")
        assert dy.shape[0] == x.shape[0]
        M, N, K = dy.shape[0], dy.shape[1], x.shape[1]
        dw = torch.empty((N, K), dtype=torch.float32, device=self.device)
        ws = torch.empty(int(self.raw_lib.a3d_wgrad_ws_floats(M, N, K)), dtype=torch.float32, device=self.device)
        _check(self.lib.a3d_wgrad_bf16(self._stream(), _p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), K, _p(ws), M, N, K, alpha, 0),
               f"a3d_wgrad_bf16 M={M} N={N} K={K}")
        return dw

    def colsum(self, x, alpha: float = 1.0):
        x = self._act(x, "colsum.x")
        out = torch.empty(x.shape[1], dtype=torch.float32, device=self.device)
        _check(self.lib.a3d_colsum_bf16(self._stream(), _p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), alpha, 0), "a3d_colsum_bf16")
        return out

    def axpby_(self, x, y, a: float = 1.0, b: float = 1.0):
        """y <- a x + b y (in place on ``y``)."""
        x, y = self._act(x, "axpby.x"), self._act(y, "axpby.y")
        assert x.is_contiguous() and y.is_contiguous() and x.shape == y.shape
        _check(self.lib.a3d_axpby_bf16(self._stream(), _p(x), _p(y), x.numel(), a, b), "a3d_axpby_bf16")
        return y

    def scaled(self, x, a: float):
        return self.axpby_(x, torch.empty_like(x), a, 0.0)

    def zero_insert2x(self, dy, B: int, H: int, W: int):
        dy = self._act(dy, "zero_insert.dy")
        assert dy.is_contiguous() and dy.shape[0] == B * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)
        z = self.empty(B * H * W, dy.shape[1])
        _check(self.lib.a3d_zero_insert2x_bf16(self._stream(), _p(dy), _p(z), B, H, W, dy.shape[1]), "a3d_zero_insert2x_bf16")
        return z

    def upsample2x_bwd(self, du, B: int, H: int, W: int, He: int, We: int):
        du = self._act(du, "upsample_bwd.du")
        assert du.is_contiguous() and du.shape[0] == B * He * We
        dx = self.empty(B * H * W, du.shape[1])
        _check(self.lib.a3d_upsample2x_bwd_bf16(self._stream(), _p(du), _p(dx), B, H, W, He, We, du.shape[1]), "a3d_upsample2x_bwd_bf16")
        return dx

    def sqnorm(self, g: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False):
        assert g.dtype == torch.float32 and g.is_contiguous()
        out = out if out is not None else torch.empty(1, dtype=torch.float32, device=self.device)
        _check(self.raw_lib.a3d_sqnorm_f32(self._stream(), _p(g), g.numel(), _p(out), 1 if accumulate else 0), "a3d_sqnorm_f32")
        return out

    def clip_ctrl(self, sq: torch.Tensor, max_norm: float, inv_loss_scale: float = 1.0):
        ctrl = torch.empty(3, dtype=torch.float32, device=self.device)
        _check(self.raw_lib.a3d_clip_ctrl_f32(self._stream(), _p(sq), max_norm, inv_loss_scale, _p(ctrl)), "a3d_clip_ctrl_f32")
        return ctrl

    def adamw_(self, p, g, m, v, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, step: int = 1, ctrl=None):
        for t in (p, g, m, v):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
        b1, b2 = betas
        _check(self.raw_lib.a3d_adamw_f32(self._stream(), _p(p), _p(g), _p(m), _p(v), p.numel(), lr, b1, b2, eps, weight_decay,
                                          1.0 - b1 ** step, 1.0 - b2 ** step, _p(ctrl)), "a3d_adamw_f32")
        return p

    def cfg_ddim_step(self, eps_pair, x, first_frame, guidance: float, alpha_t: float, alpha_prev: float):
        """Fused pipeline epilogue (pipeline.py:1023-1031); fp32 [n, C, F, H, W] tensors."""
        for t in (eps_pair, x, first_frame):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        n, C, F, H, W = x.shape
        assert eps_pair.shape == (2 * n, C, F, H, W) and first_frame.numel() == n * C * H * W
        y = torch.empty_like(x)
        _check(self.lib.a3d_cfg_ddim_step_f32(self._stream(), _p(eps_pair), _p(x), _p(first_frame), _p(y), n, C, F, H * W,
                                              guidance, alpha_t, alpha_prev), "a3d_cfg_ddim_step_f32")
        return y
