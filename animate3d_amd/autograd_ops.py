"""Differentiable view of the op set (training path).

The reference trains the motion modules and the ``*_i2v`` projections with plain torch autograd
(``train.py:576-590``: ``unet(...)`` under autocast, ``loss.backward()``).  The drop-in keeps that contract:
``AutogradOps(base)`` exposes the same methods as ``HipOps`` (``animate3d_amd/unet.py`` is written against them), each one a
``torch.autograd.Function`` whose forward is the inference kernel and whose backward is built from the backward kernels of
``include/animate3d_hip.h`` ("Training path") plus the forward GEMM / conv kernels on transposed operands:

    gemm            dX = dY W            (a3d_gemm on W^T),  dW = dY^T X (a3d_wgrad: split over the token axis),  db = a3d_colsum
    conv3x3         dX = conv(dY, flipped W^T) (+ a3d_zero_insert2x for stride 2, a3d_upsample2x_bwd behind the up-sampler)
    gemm_geglu      projection recomputed, a3d_geglu_bwd, then as gemm
    flash_attn      a3d_flash_attn_bwd     temporal_attn  a3d_temporal_attn_bwd
    group_norm      a3d_group_norm_sums + a3d_group_norm_bwd        layer_norm  a3d_layer_norm_bwd

torch's autograd engine only walks the graph (and sums the gradients of tensors that feed several branches); no arithmetic of
the backward pass is left to eager torch except those sums, the column slices of fused projections and the merge-coefficient
reduction over a weight-sized tensor (see ``_Gemm.backward``).  ``base`` is ``HipOps`` in the product; the CPU tests run the same
class over ``tests/torch_ops.TorchRefOps`` to check every formula against autograd of the oracle without a GPU.
"""
from __future__ import annotations

import contextlib
from typing import Optional

import torch

from .hip_ops import RowMap


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.contiguous()


class _GradSink:
    """fp32 gradient of a packed kernel weight, handed from the GEMM backward that computed it (``a3d_wgrad`` produces fp32) to ``PackW.backward``
    next to the 16-bit placeholder autograd carries between the two: no fp32 -> 16-bit -> fp32 round trip of every weight gradient."""
    __slots__ = ("grad32", "announced")

    def __init__(self):
        self.grad32 = None
        self.announced = False       # one placeholder per backward pass has been handed to autograd (see _weight_grad)

    def put(self, g: torch.Tensor) -> None:
        self.grad32 = g if self.grad32 is None else self.grad32 + g


class _SinkRows:
    """The sink of a row range of a packed weight (``weight_rows``): the consumer of ``w_kvq[:2 * C]`` deposits its fp32 gradient into
    those rows of the parent's sink instead of sending a 16-bit gradient through autograd's slice backward."""
    __slots__ = ("parent", "r0", "rows_total")

    def __init__(self, parent, r0: int, rows_total: int):
        self.parent, self.r0, self.rows_total = parent, r0, rows_total

    @property
    def announced(self):
        return self.parent.announced

    @announced.setter
    def announced(self, v):
        self.parent.announced = v

    def put(self, g: torch.Tensor) -> None:
        p = self.parent
        if p.grad32 is None:
            p.grad32 = torch.zeros((self.rows_total, *g.shape[1:]), dtype=torch.float32, device=g.device)
        p.grad32[self.r0:self.r0 + g.shape[0]] += g


_PLACEHOLDERS = {}


class _Deferred:
    """Weight gradients parked by ``PackW.backward`` while ``deferred_param_grads()`` is active.  Parked slices keep their fused fp32
    gradient buffers alive (AccumulateGrad would have consumed and freed each as it arrived), so the list is flushed — one multi-tensor add —
    whenever more than ``FLUSH_BYTES`` are parked: the extra peak memory is bounded by that, not by one fp32 copy of every trainable gradient."""
    active = False
    dst: list = []
    src: list = []
    parked_bytes = 0
    FLUSH_BYTES = 256 << 20

    @classmethod
    def park(cls, dst, src):
        cls.dst.append(dst)
        cls.src.append(src)
        cls.parked_bytes += src.numel() * src.element_size()
        if cls.parked_bytes > cls.FLUSH_BYTES:
            cls.flush()

    @classmethod
    def flush(cls):
        dst, src, cls.dst, cls.src, cls.parked_bytes = cls.dst, cls.src, [], [], 0
        if dst:
            with torch.no_grad():
                torch._foreach_add_(dst, src)


@contextlib.contextmanager
def deferred_param_grads():
    """Around ``loss.backward()`` of a loop that owns its gradient buffers (``train.training_step`` with ``FlatAdamW``): the fp32 gradient
    slices that ``PackW.backward`` would hand to autograd's AccumulateGrad nodes — one small ``grad.add_`` launch per trainable weight, ~850 per
    step in the reference's configuration, launch-bound — are parked and added to the parameters' ``.grad`` in ONE multi-tensor call when
    the backward pass is over.  Not for loops that hang hooks on gradient accumulation (torch DDP): leave those on the plain path."""
    st = _Deferred
    prev, st.active = st.active, True
    try:
        yield
    except BaseException:
        st.dst, st.src, st.parked_bytes = [], [], 0
        raise
    finally:
        st.active = prev
    st.flush()


def _placeholder(w: torch.Tensor) -> torch.Tensor:
    """A zero of w's dtype expanded to its shape (no memory, no kernel): keeps autograd's edge to PackW alive while the real gradient travels in the sink."""
    key = (w.dtype, w.device)
    z = _PLACEHOLDERS.get(key)
    if z is None:
        z = _PLACEHOLDERS[key] = torch.zeros((), dtype=w.dtype, device=w.device)
    return z.expand(w.shape)


class PackW(torch.autograd.Function):
    """Trainable fp32 master weights -> the 16-bit kernel operand of one GEMM: rows concatenated (fused Q|K|V projections), optionally
    GEGLU-interleaved, cast to the storage type.  The per-step price of fp32 master weights behind 16-bit kernels (unet._pack_train);
    backward splits the operand's gradient back over the masters — in fp32, taken from the sink when the consuming GEMM left it there."""

    @staticmethod
    def forward(ctx, dtype, interleave, sink, *masters):
        ctx.interleave, ctx.sink = interleave, sink
        ctx.rows = [m.shape[0] for m in masters]
        ctx.save_for_backward(*masters)          # (not pinned on ctx: the graph frees them with its other saved tensors)
        if len(masters) == 1:
            w = masters[0].to(dtype)
        else:                        # cast while concatenating: no fp32 copy of the fused operand in between
            w = torch.empty((sum(ctx.rows), *masters[0].shape[1:]), dtype=dtype, device=masters[0].device)
            r = 0
            for m in masters:
                w[r:r + m.shape[0]].copy_(m)
                r += m.shape[0]
        if interleave:               # a row permutation: the same bits whether it runs before or after the cast
            w = _interleave32(w)
        return w.contiguous()

    @staticmethod
    def backward(ctx, dw):
        g, ctx.sink.grad32 = ctx.sink.grad32, None
        announced, ctx.sink.announced = ctx.sink.announced, False
        # Sink consumers hand autograd exactly ONE stride-0 zero placeholder per backward pass (the first of them; the others return
        # None — see _weight_grad), so ``dw`` is that placeholder alone, or a dense tensor that contains a real gradient: a consumer
        # of a plain slice / view of the packed operand that went around the sink (``weight_rows`` gives slices a sink of their own).
        real = dw is not None and not (announced and dw.dim() > 0 and all(s_ == 0 for s_ in dw.stride()))
        if g is None:
            g = dw.float()
        elif real:
            g = g + dw.float()
        if ctx.interleave:
            g = _deinterleave32(g)
        outs, r = [], 0
        for i, n in enumerate(ctx.rows):
            gi = g[r:r + n] if ctx.needs_input_grad[3 + i] else None
            if gi is not None and _Deferred.active:
                m = ctx.saved_tensors[i]
                if m.is_leaf and m.grad is not None and m.grad.dtype == gi.dtype and m.grad.shape == gi.shape:
                    _Deferred.park(m.grad, gi)            # deferred_param_grads(): added in multi-tensor launches, not one per weight
                    gi = None
            outs.append(gi)
            r += n
        return (None, None, None, *outs)


def _interleave32(w):          # hip_ops.HipOps.interleave_geglu: [h rows | gate rows] -> blocks of 32 rows alternating
    n = w.shape[0] // 2
    return torch.stack([w[:n].reshape(n // 32, 32, *w.shape[1:]), w[n:].reshape(n // 32, 32, *w.shape[1:])], dim=1).reshape(w.shape)


def _deinterleave32(w):
    n = w.shape[0] // 2
    return w.reshape(n // 32, 2, 32, *w.shape[1:]).transpose(0, 1).reshape(w.shape)          # one copy: [h rows | gate rows]


def pack_weight(dtype, masters, interleave: bool = False) -> torch.Tensor:
    sink = _GradSink()
    w = PackW.apply(dtype, interleave, sink, *masters)
    w._a3d_sink = sink
    return w


def _param_grad(p, g):
    """Gradient of a bias / affine vector as a backward returns it: parked under ``deferred_param_grads()`` when ``p`` is the fp32 parameter
    itself (its ``.grad`` then takes it in the multi-tensor add after the pass), handed to autograd otherwise."""
    if (g is not None and _Deferred.active and p is not None and p.is_leaf and p.requires_grad and p.grad is not None
            and p.grad.dtype == g.dtype and p.grad.shape == g.shape):
        _Deferred.park(p.grad, g)
        return None
    return g


def _weight_grad(ctx_sink, w, dw32: torch.Tensor) -> torch.Tensor:
    """What a GEMM backward returns for its weight operand: through the sink (fp32, no cast) when the operand came from ``pack_weight``."""
    if ctx_sink is not None:
        ctx_sink.put(dw32)
        if ctx_sink.announced:          # autograd already holds this pass's placeholder: nothing more to sum into it
            return None
        ctx_sink.announced = True
        return _placeholder(w)
    return dw32.to(w.dtype)


class _WeightRows(torch.autograd.Function):
    """``w[r0:r1]`` whose backward keeps the sink consumers' stride-0 zero placeholder memory-free: autograd's own slice backward would
    materialise it as a dense full-size zero tensor, which ``PackW.backward`` then takes for a real gradient (a full-width fp32 add per pass:
    w_kvq and qkv_img in the frame-sharded and first-frame paths)."""

    @staticmethod
    def forward(ctx, w, r0, r1):
        ctx.shape, ctx.r0, ctx.r1 = tuple(w.shape), r0, r1
        ctx.set_materialize_grads(False)
        return w[r0:r1]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        if g.dim() > 0 and all(s_ == 0 for s_ in g.stride()):      # the placeholder of _weight_grad: pass one of the parent's shape on
            key = (g.dtype, g.device)
            z = _PLACEHOLDERS.get(key)
            if z is None:
                z = _PLACEHOLDERS[key] = torch.zeros((), dtype=g.dtype, device=g.device)
            return z.expand(ctx.shape), None, None
        full = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)      # a consumer that went around the sink: a real gradient
        full[ctx.r0:ctx.r1] = g
        return full, None, None


def weight_rows(w: torch.Tensor, r0: int, r1: int) -> torch.Tensor:
    """``w[r0:r1]`` of a packed kernel weight that keeps the fp32 gradient path: the slice carries a sink that deposits into rows
    [r0, r1) of the parent's (plain slicing drops ``_a3d_sink``, and the consumer's weight gradient would travel and be summed in 16 bits)."""
    sink = getattr(w, "_a3d_sink", None)
    v = _WeightRows.apply(w, r0, r1) if (sink is not None and w.requires_grad) else w[r0:r1]
    if sink is not None:
        v._a3d_sink = _SinkRows(sink, r0, w.shape[0])
    return v


class _GradCols:
    """The gradient of a fused projection output [rows, total] whose column pieces feed different kernels (K | V | Q | Q_i2v of one GEMM):
    ONE buffer, allocated by the first backward that writes a piece.  An attention backward that is the only consumer of a piece writes
    its dQ / dK / dV straight into that piece's columns (``hip_ops.flash_attn_bwd(dq_out=...)``); ``_SplitCols.backward`` copies in whatever
    arrives as a separate tensor (a piece with several consumers: autograd has summed their gradients) and zeroes pieces nobody used.
    Plain slicing costs, per piece, a zero-filled full-width tensor plus a copy, and a full-width add per extra piece."""
    __slots__ = ("base", "rows", "bounds", "uses", "buf")

    def __init__(self, base, rows: int, bounds):
        self.base, self.rows, self.bounds = base, rows, tuple(bounds)
        self.uses = [0] * (len(bounds) - 1)          # consumers that can write in place (counted in their forward)
        self.buf = None

    def piece(self, i: int, buf=None) -> torch.Tensor:
        if buf is None:
            if self.buf is None:
                self.buf = self.base.empty(self.rows, self.bounds[-1])
            buf = self.buf
        return buf[:, self.bounds[i]:self.bounds[i + 1]]


class _SplitCols(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hold, x):
        ctx.hold = hold
        ctx.set_materialize_grads(False)
        b = hold.bounds
        return tuple(x[:, b[i]:b[i + 1]] for i in range(len(b) - 1))

    @staticmethod
    def backward(ctx, *gs):
        hold = ctx.hold
        buf, hold.buf = hold.buf, None
        if buf is None:
            if all(g is None for g in gs):
                return None, None
            buf = hold.base.empty(hold.rows, hold.bounds[-1])
        for i, g in enumerate(gs):
            dst = hold.piece(i, buf)
            if g is None:
                dst.zero_()
            elif not (g.data_ptr() == dst.data_ptr() and g.shape == dst.shape and g.stride() == dst.stride()):
                dst.copy_(g)             # (else: written in place by the piece's only consumer)
        return None, buf


class _Gemm(torch.autograd.Function):
    """Y = alpha (X W^T + bias + rowbias) + beta R;  ``alpha_t``: optional 0-dim tensor the float ``alpha`` was read from (the
    AlphaBlender merge weight of attention_processor.py:700-713, a trainable scalar)."""

    @staticmethod
    def forward(ctx, aops, x, w, bias, residual, rowbias, alpha_t, alpha, beta, rb_div):
        base = aops.base
        ctx.aops, ctx.alpha, ctx.beta = aops, alpha, beta
        ctx.has_rowbias = rowbias is not None
        ctx.w_sink = getattr(w, "_a3d_sink", None)
        ctx.save_for_backward(x, w, bias)
        return base.gemm(x, w, bias, residual=residual, alpha=alpha, beta=beta, rowbias=rowbias, rb_div=rb_div)

    @staticmethod
    def backward(ctx, dy):
        aops, base = ctx.aops, ctx.aops.base
        x, w, bias = ctx.saved_tensors
        need = ctx.needs_input_grad
        if need[5]:
            raise NotImplementedError("gradient of a GEMM row-bias (time-embedding projection) is not part of the training path")
        if need[6] and ctx.has_rowbias:
            raise NotImplementedError("gradient of a tensor merge weight on a GEMM that also carries a row-bias (d alpha would need the row-bias term)")
        dx = dw = db = dres = dalpha = None
        if need[1]:
            dx = base.gemm(dy, aops.transposed_weight(w), alpha=ctx.alpha)
        if need[6]:
            # d alpha = sum dY * (X W^T + bias) = sum W * (dY^T X) + bias . colsum(dY): the fp32 weight gradient, then a weight-sized
            # reduction — heavy cancellation, so the 16-bit rounding of dW must not come first
            dw_u = base.wgrad(dy, x)                                                # dY^T X  [N, K] fp32
            db_u = base.colsum(dy) if bias is not None else None
            dalpha = (w.float() * dw_u).sum()
            if db_u is not None:
                dalpha = dalpha + (bias.float() * db_u).sum()
            if need[2]:
                dw = _weight_grad(ctx.w_sink, w, dw_u * ctx.alpha)
            if need[3]:
                db = db_u * ctx.alpha
        elif need[2]:
            dw = _weight_grad(ctx.w_sink, w, base.wgrad(dy, x, ctx.alpha))
            if need[3]:
                db = base.colsum(dy, ctx.alpha)
        elif need[3]:
            db = base.colsum(dy, ctx.alpha)
        if need[4]:
            dres = dy if ctx.beta == 1.0 else base.scaled(_c(dy), ctx.beta)
        return None, dx, dw, _param_grad(bias, db), dres, None, dalpha, None, None, None


class _GemmGeglu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aops, x, w_il, b_il):
        ctx.aops = aops
        ctx.w_sink = getattr(w_il, "_a3d_sink", None)
        ctx.save_for_backward(x, w_il, b_il)
        return aops.base.gemm_geglu(x, w_il, b_il)

    @staticmethod
    def backward(ctx, dy):
        aops, base = ctx.aops, ctx.aops.base
        x, w_il, b_il = ctx.saved_tensors
        need = ctx.needs_input_grad
        proj = base.gemm(x, w_il, b_il)                     # recomputed: the forward keeps only its input
        dp = base.geglu_bwd(proj, _c(dy))
        dx = base.gemm(dp, aops.transposed_weight(w_il)) if need[1] else None
        dw = _weight_grad(ctx.w_sink, w_il, base.wgrad(dp, x)) if need[2] else None
        db = base.colsum(dp) if need[3] else None
        return None, dx, dw, _param_grad(b_il, db)


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aops, x, w, bias, residual, rowbias, B, H, W, stride, up2x, rb_div, up_size):
        ctx.aops, ctx.geom = aops, (B, H, W, stride, up2x, up_size)
        ctx.save_for_backward(w)
        ctx.cin = x.shape[1]
        y, _, _ = aops.base.conv3x3(x, B, H, W, w, bias, stride=stride, up2x=up2x, rowbias=rowbias, rb_div=rb_div, residual=residual, up_size=up_size)
        return y

    @staticmethod
    def backward(ctx, dy):
        aops, base = ctx.aops, ctx.aops.base
        (w,) = ctx.saved_tensors
        B, H, W, stride, up2x, up_size = ctx.geom
        need = ctx.needs_input_grad
        if need[2] or need[3] or need[5]:
            raise NotImplementedError("no 3x3 convolution is trainable in the reference's configuration (train.yaml: trainable_modules = "
                                      "'i2v.', 'motion_modules.'); weight / bias / row-bias gradients of a3d_conv3x3 are not built")
        dy = _c(dy)
        dx = None
        if need[1]:
            if dy.shape[1] % 8 != 0:          # conv_out: 4 output channels -> the im2col + K=64 GEMM route of conv_in
                assert stride == 1 and not up2x
                img = dy.reshape(B, 1, H, W, dy.shape[1]).permute(0, 4, 1, 2, 3).contiguous()
                dx = base.gemm(base.im2col_in(img), aops.dgrad_weight_small(w, ctx.cin))
            else:
                wd = aops.dgrad_weight(w, ctx.cin)
                if up2x:
                    He, We = (2 * H, 2 * W) if up_size is None else (int(up_size[0]), int(up_size[1]))
                    du, _, _ = base.conv3x3(dy, B, He, We, wd, None)
                    dx = base.upsample2x_bwd(du, B, H, W, He, We)
                elif stride == 2:
                    dx, _, _ = base.conv3x3(base.zero_insert2x(dy, B, H, W), B, H, W, wd, None)
                else:
                    dx, _, _ = base.conv3x3(dy, B, H, W, wd, None)
        dres = dy if need[4] else None
        return None, dx, None, None, dres, None, None, None, None, None, None, None, None


class _FlashAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aops, q, k, v, out_in, qmap, kmap, groups, heads, q_len, kv_len, out_scale, may_keep_out=True):
        ctx.aops, ctx.args = aops, (qmap, kmap, groups, heads, q_len, kv_len, out_scale)
        ctx.pieces = tuple(getattr(t, "_a3d_gcols", None) for t in (q, k, v))
        for tag in ctx.pieces:
            if tag is not None:
                tag[0].uses[tag[1]] += 1
        if out_in is None:
            # the forward hands its log-sum-exp to the backward (and autograd keeps the output anyway, as the next GEMM's input): the
            # backward's statistics pass — a third of its time at head_dim 40 — is not run
            if aops.keep_lse and may_keep_out:
                o, lse = aops.base.flash_attn(q, k, v, qmap, kmap, groups, heads, q_len, kv_len, out_scale=out_scale, with_lse=True)
                ctx.save_for_backward(q, k, v, o, lse)
                o._a3d_kept_for_backward = True
                return o
            ctx.save_for_backward(q, k, v)
            return aops.base.flash_attn(q, k, v, qmap, kmap, groups, heads, q_len, kv_len, out_scale=out_scale)
        # accumulated into a caller's buffer (the IP-Adapter image tokens on top of the text attention): that buffer is not this
        # attention's output, the backward recomputes the statistics
        if getattr(out_in, "_a3d_kept_for_backward", False):
            raise RuntimeError("flash_attn(out=..., accumulate=True) adds into the result of a flash_attn call that kept that result for its "
                               "backward (delta = rowsum(dO * O)): pass accumulation_target=True to the call that produced the buffer")
        ctx.save_for_backward(q, k, v)
        aops.base.flash_attn(q, k, v, qmap, kmap, groups, heads, q_len, kv_len, out=out_in, out_scale=out_scale, accumulate=True)
        ctx.mark_dirty(out_in)
        return out_in

    @staticmethod
    def backward(ctx, dy):
        base = ctx.aops.base
        q, k, v, *kept = ctx.saved_tensors
        o, lse = kept if kept else (None, None)
        qmap, kmap, groups, heads, q_len, kv_len, out_scale = ctx.args
        need = ctx.needs_input_grad
        dq = dk = dv = None
        if need[1] or need[2] or need[3]:
            # query groups that differ only in g % gdiv read the same K/V rows when the map ignores that index (gb == 0):
            # the F frames of a video in the first-frame branches
            g_share = kmap.gdiv if (kmap.gb == 0 and kmap.gdiv > 1 and groups % kmap.gdiv == 0) else 1
            # pieces of a fused projection output that only this attention reads: their gradient is written into the columns of the
            # projection output's gradient buffer (_GradCols) instead of a tensor of its own
            inplace = {}
            pq, pk_, pv_ = (tag if tag is not None and tag[0].uses[tag[1]] == 1 else None for tag in ctx.pieces)
            if pq is not None and need[1]:
                inplace["dq_out"] = pq[0].piece(pq[1])
            if pk_ is not None and pv_ is not None and (need[2] or need[3]):
                inplace["dk_out"], inplace["dv_out"] = pk_[0].piece(pk_[1]), pv_[0].piece(pv_[1])
            dq, dk, dv = base.flash_attn_bwd(q, k, v, _c(dy), qmap, kmap, groups, heads, q_len, kv_len, q_per_kv=g_share, do_scale=out_scale,
                                             need_dq=need[1], need_dkv=need[2] or need[3], **({} if lse is None else dict(o=o, lse=lse)), **inplace)
        return None, dq, dk, dv, (dy if need[4] else None), None, None, None, None, None, None, None, None


class _TemporalAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aops, q, k, v, videos, frames, L, heads):
        ctx.aops, ctx.args = aops, (videos, frames, L, heads)
        ctx.save_for_backward(q, k, v)
        return aops.base.temporal_attn(q, k, v, videos, frames, L, heads)

    @staticmethod
    def backward(ctx, dy):
        q, k, v = ctx.saved_tensors
        C = q.shape[1]
        d = ctx.aops.base.temporal_attn_bwd(q, k, v, _c(dy), *ctx.args)
        return None, d[:, :C], d[:, C:2 * C], d[:, 2 * C:], None, None, None, None


class _TemporalAttnFused(torch.autograd.Function):
    """Same, with Q | K | V the three column ranges of ONE projection output: the gradient is written as one [rows, 3C] buffer."""

    @staticmethod
    def forward(ctx, aops, qkv, videos, frames, L, heads):
        ctx.aops, ctx.args = aops, (videos, frames, L, heads)
        ctx.save_for_backward(qkv)
        C = qkv.shape[1] // 3
        return aops.base.temporal_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], videos, frames, L, heads)

    @staticmethod
    def backward(ctx, dy):
        (qkv,) = ctx.saved_tensors
        C = qkv.shape[1] // 3
        return None, ctx.aops.base.temporal_attn_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], _c(dy), *ctx.args), None, None, None, None


class _GroupNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aops, x, gamma, beta, B, rows, groups, eps, silu):
        ctx.aops, ctx.args = aops, (B, rows, groups, eps, silu)
        ctx.save_for_backward(x, gamma, beta)
        return aops.base.group_norm(x, B, rows, gamma, beta, groups, eps, silu)

    @staticmethod
    def backward(ctx, dy):
        base = ctx.aops.base
        x, gamma, beta = ctx.saved_tensors
        B, rows, groups, eps, silu = ctx.args
        need = ctx.needs_input_grad
        stats = base.group_norm_stats(x, B, rows, groups, eps)
        dx, dg, db = base.group_norm_bwd(x, _c(dy), B, rows, gamma, beta, groups, stats, silu, need_param=need[2] or need[3])
        return (None, (dx if need[1] else None), _param_grad(gamma, dg if need[2] else None), _param_grad(beta, db if need[3] else None),
                None, None, None, None, None)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aops, x, gamma, beta, eps, pe1, pe1_div, pe2, pe2_div, two):
        ctx.aops, ctx.eps = aops, eps
        ctx.save_for_backward(x, gamma, beta)
        ctx.set_materialize_grads(False)
        out = aops.base.layer_norm(x, gamma, beta, eps, pe1=pe1, pe1_div=pe1_div, pe2=pe2, pe2_div=pe2_div, two=two)
        return out if two else out

    @staticmethod
    def backward(ctx, *dys):
        base = ctx.aops.base
        x, gamma, beta = ctx.saved_tensors
        need = ctx.needs_input_grad
        dys = [d for d in dys if d is not None]
        if not dys:
            return (None,) * 10
        d = _c(dys[0])
        if len(dys) > 1:                     # both outputs of a two-encoding call were used: their gradients add (fp32 sum, one rounding)
            d = d + dys[1]
        dx, dg, db = base.layer_norm_bwd(x, d, gamma, ctx.eps, need_param=need[2] or need[3])
        return (None, (dx if need[1] else None), _param_grad(gamma, dg if need[2] else None), _param_grad(beta, db if need[3] else None),
                None, None, None, None, None, None)


class _Concat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aops, a, b):
        ctx.ca = a.shape[1]
        return aops.base.concat(a, b)

    @staticmethod
    def backward(ctx, dy):
        return None, dy[:, :ctx.ca], dy[:, ctx.ca:]


class _UnpackOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aops, x, V, C, F, H, W, dtype):
        ctx.dt = x.dtype
        return aops.base.unpack_out(x, V, C, F, H, W, dtype)

    @staticmethod
    def backward(ctx, dout):      # [V, C, F, H, W] -> rows ((v f) h w), C: a layout change of a 4-channel tensor
        return None, dout.permute(0, 2, 3, 4, 1).reshape(-1, dout.shape[1]).to(ctx.dt).contiguous(), None, None, None, None, None, None


class AutogradOps:
    """``HipOps``-compatible op set whose results carry autograd history.  Anything not listed here (``empty``, ``act_dtype``,
    ``timestep_embed``, ``im2col_in`` ... — ops whose inputs never require a gradient) is forwarded to ``base`` unchanged."""

    def __init__(self, base):
        self.base = base
        self._persistent = {}       # data_ptr -> (weight, derived operand): frozen weights of the persistent pack only
        self.keep_lse = bool(getattr(base, "has_attn_lse", False))     # forward log-sum-exp -> backward (HipOps; the torch reference op set has none)
        self._host_scalars = {}     # id(0-dim tensor) -> (tensor, python float): merge weights read back in ONE transfer per step

    flash_attn2 = None       # no differentiable fused text + image-token attention: the UNet issues the two differentiable calls

    def __getattr__(self, name):
        return getattr(self.base, name)

    # ---- derived weight operands of the dgrad products
    def _cached(self, w, tag, make):
        if getattr(w, "_a3d_persistent", False) and not w.requires_grad:
            key = (w.data_ptr(), tuple(w.shape), tag)
            hit = self._persistent.get(key)
            if hit is None:
                hit = (w, make())
                self._persistent[key] = hit
            return hit[1]
        return make()

    def transposed_weight(self, w):
        """W [N, K] -> W^T [K, N] (the ``weight`` operand of dX = dY W)."""
        return self._cached(w, "t", lambda: self.base.transpose(w.detach(), pad=1))

    def dgrad_weight(self, w, cin: int):
        """Packed conv weight [Cout, (ky, kx, ci)] -> [Cin, (2-ky, 2-kx, co)]: the same kernel computes the input gradient."""
        def make():
            cout = w.shape[0]
            return w.detach().reshape(cout, 3, 3, cin).flip(1, 2).permute(3, 1, 2, 0).reshape(cin, 9 * cout).contiguous()
        return self._cached(w, "d", make)

    def dgrad_weight_small(self, w, cin: int):
        """Same for a conv with < 8 output channels (conv_out): [Cin, 64] over im2col patches of the output gradient."""
        def make():
            cout = w.shape[0]
            wd = w.detach().reshape(cout, 3, 3, cin).flip(1, 2).permute(3, 1, 2, 0).reshape(cin, 9 * cout)
            out = torch.zeros((cin, 64), dtype=w.dtype, device=w.device)
            out[:, : 9 * cout] = wd
            return out
        return self._cached(w, "ds", make)

    def prefetch_scalars(self, tensors):
        """The kernels take merge weights by value (``a3d_gemm``'s ``alpha``): a trainable weight (AlphaBlender ``mix_factor``) has to be read
        back from the device.  One ``float(t)`` per GEMM is a host synchronisation in the middle of the forward (~130 per training step:
        the launch queue drains every time); this reads all of them in one transfer before the forward starts."""
        ts = [t for t in tensors if torch.is_tensor(t)]
        self._host_scalars = {}
        if ts:
            vals = torch.stack([t.detach().float().reshape(()) for t in ts]).tolist()
            self._host_scalars = {id(t): (t, v) for t, v in zip(ts, vals)}       # (the tensor is kept alive so that its id stays its own)

    def _scalar(self, t) -> float:
        hit = self._host_scalars.get(id(t))
        return hit[1] if hit is not None and hit[0] is t else float(t.detach())

    # ---- the op set
    def gemm(self, x, w, bias=None, *, residual=None, alpha=1.0, beta: float = 1.0, rowbias=None, rb_div: int = 1, out=None):
        if out is not None:
            raise NotImplementedError("gemm(out=...) has no autograd form")
        alpha_t = alpha if torch.is_tensor(alpha) else None
        return _Gemm.apply(self, x, w, bias, residual, rowbias, alpha_t, self._scalar(alpha) if alpha_t is not None else float(alpha), beta, rb_div)

    def split_cols(self, x, *bounds):
        """Column pieces [bounds[i], bounds[i+1]) of a fused projection output (``bounds`` from 0 to its width) whose gradients meet again
        in ONE buffer (_GradCols) instead of one zero-padded full-width tensor per piece summed by autograd."""
        if not (torch.is_tensor(x) and x.requires_grad and torch.is_grad_enabled()) or bounds[0] != 0 or bounds[-1] != x.shape[1]:
            return tuple(x[:, a:b] for a, b in zip(bounds[:-1], bounds[1:]))
        hold = _GradCols(self.base, x.shape[0], bounds)
        outs = _SplitCols.apply(hold, x)
        for i, o in enumerate(outs):
            o._a3d_gcols = (hold, i)
        return outs

    def gemm_geglu(self, x, w_il, bias_il):
        return _GemmGeglu.apply(self, x, w_il, bias_il)

    def conv3x3(self, x, B, H, W, w, bias, *, stride: int = 1, up2x: bool = False, rowbias=None, rb_div: int = 1, residual=None, up_size=None):
        y = _Conv3x3.apply(self, x, w, bias, residual, rowbias, B, H, W, stride, up2x, rb_div, up_size)
        He, We = ((2 * H, 2 * W) if up_size is None else (int(up_size[0]), int(up_size[1]))) if up2x else (H, W)
        return y, (He - 1) // stride + 1, (We - 1) // stride + 1

    def flash_attn(self, q, k, v, qmap: RowMap, kmap: RowMap, groups: int, heads: int, q_len: int, kv_len: int, *,
                   out=None, out_scale: float = 1.0, accumulate: bool = False, causal: bool = False, accumulation_target: bool = False):
        """``accumulation_target``: a later call will add into this call's result in place (the IP-Adapter attention on top of the text
        attention): the result then is not this attention's output any more and must not be kept for the backward."""
        if causal:
            raise NotImplementedError("causal attention (CLIP text tower) is not on the training path")
        if (out is None) != (not accumulate):
            raise NotImplementedError("flash_attn(out=...) is differentiable only as an accumulation into an existing result")
        return _FlashAttn.apply(self, q, k, v, out, qmap, kmap, groups, heads, q_len, kv_len, out_scale, not accumulation_target)

    def temporal_attn(self, q, k, v, videos: int, frames: int, L: int, heads: int, *, q_f0: int = 0, q_frames=None):
        if q_frames is not None and q_frames != frames:
            raise NotImplementedError("the frame-sharded temporal attention has no backward (training shards the batch, not the frames)")
        base = q._base
        C = q.shape[1]
        if (base is not None and k._base is base and v._base is base and base.dim() == 2 and base.shape[1] == 3 * C and base.is_contiguous()
                and q.storage_offset() == base.storage_offset() and k.storage_offset() == base.storage_offset() + C
                and v.storage_offset() == base.storage_offset() + 2 * C and q.stride() == k.stride() == v.stride() == base.stride()):
            return _TemporalAttnFused.apply(self, base, videos, frames, L, heads)
        return _TemporalAttn.apply(self, q, k, v, videos, frames, L, heads)

    def group_norm(self, x, B, rows, gamma, beta, groups, eps, silu):
        return _GroupNorm.apply(self, x, gamma, beta, B, rows, groups, eps, silu)

    def layer_norm(self, x, gamma, beta, eps, pe1=None, pe1_div: int = 1, pe2=None, pe2_div: int = 1, two: bool = False):
        return _LayerNorm.apply(self, x, gamma, beta, eps, pe1, pe1_div, pe2, pe2_div, two)

    def concat(self, a, b):
        return _Concat.apply(self, a, b)

    def unpack_out(self, x, V, C, F, H, W, dtype):
        return _UnpackOut.apply(self, x, V, C, F, H, W, dtype)

    def _no_grad_op(self, name, x, *args):
        if torch.is_tensor(x) and x.requires_grad:
            raise NotImplementedError(f"{name} has no backward on the training path (its input never requires a gradient in the reference's "
                                      "configuration: the time / camera embeddings are frozen)")
        return getattr(self.base, name)(x, *args)

    def silu(self, x):
        return self._no_grad_op("silu", x)

    def activation(self, x, kind):
        return self._no_grad_op("activation", x, kind)

    def geglu(self, x):
        return self._no_grad_op("geglu", x)

    def group_norm_apply(self, *a, **k):
        raise NotImplementedError("the rank-split GroupNorm has no backward (training shards the batch, not the frames)")

    def softmax_rows(self, *a, **k):
        raise NotImplementedError("softmax_rows has no backward on the training path")

    def gemm_f32out(self, *a, **k):
        raise NotImplementedError("gemm_f32out has no backward on the training path")
