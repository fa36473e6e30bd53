"""Plain-PyTorch reference of the op set in animate3d_amd/hip_ops.py (TEST INFRASTRUCTURE).

Same method names, argument meaning and row-major ``[rows, C]`` conventions as ``HipOps``, computed
with stock torch ops in fp32.  Used (a) as the per-op reference the HIP kernels are compared with on
the GPU and (b) to execute the product's host logic (animate3d_amd/unet.py) on a CPU so that its
layout / addressing / weight-packing logic can be checked against the oracle without a GPU.
The product never imports this file.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def rowmap_indices(m, groups: int, length: int) -> torch.Tensor:
    """[groups, length] row indices of a RowMap (include/animate3d_hip.h: a3d_rowmap)."""
    g = torch.arange(groups)[:, None]
    s = torch.arange(length)[None, :]
    return (g // m.gdiv) * m.ga + (g % m.gdiv) * m.gb + (s // m.seg_len) * m.seg_stride + (s % m.seg_len)


class TorchRefOps:
    def __init__(self, act_dtype=torch.float32, device="cpu"):
        self.act_dtype = act_dtype
        self.device = torch.device(device)

    def _o(self, t):
        return t.to(self.act_dtype)

    def split_cols(self, x, *bounds):
        return tuple(x[:, a:b] for a, b in zip(bounds[:-1], bounds[1:]))

    def empty(self, rows, cols):
        return torch.empty((rows, cols), dtype=self.act_dtype, device=self.device)

    def gemm(self, x, w, bias=None, *, residual=None, alpha=1.0, beta=1.0, rowbias=None, rb_div=1, out=None):
        y = x.float() @ w.float().t()
        if bias is not None:
            y = y + bias.float()
        if rowbias is not None:
            idx = torch.arange(x.shape[0], device=x.device) // rb_div
            y = y + rowbias.float()[idx]
        y = alpha * y
        if residual is not None:
            y = y + beta * residual.float()
        y = self._o(y)
        if out is not None:
            out.copy_(y)
            return out
        return y

    def gemm2(self, xa, xb, w, bias=None):
        return self.gemm(torch.cat([xa, xb], dim=1), w, bias)

    def group_norm2(self, xa, xb, B, rows, gamma, beta, groups, eps, silu):
        return self.group_norm(torch.cat([xa, xb], dim=1), B, rows, gamma, beta, groups, eps, silu)

    @staticmethod
    def interleave_geglu(w):
        n = w.shape[0] // 2
        h, g = w[:n].reshape(n // 32, 32, *w.shape[1:]), w[n:].reshape(n // 32, 32, *w.shape[1:])
        return torch.stack([h, g], dim=1).reshape(w.shape).contiguous()

    @staticmethod
    def deinterleave_geglu(w):
        n = w.shape[0] // 2
        blk = w.reshape(n // 32, 2, 32, *w.shape[1:])
        return torch.cat([blk[:, 0].reshape(n, *w.shape[1:]), blk[:, 1].reshape(n, *w.shape[1:])], 0)

    def gemm_geglu(self, x, w_il, bias_il):
        w = self.deinterleave_geglu(w_il.float())
        b = self.deinterleave_geglu(bias_il.float())
        h, gate = (x.float() @ w.t() + b).chunk(2, dim=-1)
        return self._o(h * F.gelu(gate))

    def conv3x3(self, x, B, H, W, w, bias, *, stride=1, up2x=False, rowbias=None, rb_div=1, residual=None, up_size=None):
        Cin, Cout = x.shape[1], w.shape[0]
        img = x.float().reshape(B, H, W, Cin).permute(0, 3, 1, 2)
        if up2x and up_size is not None:
            img = F.interpolate(img, size=tuple(up_size), mode="nearest")
        elif up2x:
            img = F.interpolate(img, scale_factor=2.0, mode="nearest")
        w4 = w.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        y = F.conv2d(img, w4, None if bias is None else bias.float(), stride=stride, padding=1)
        Ho, Wo = y.shape[2], y.shape[3]
        y = y.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout)
        if rowbias is not None:
            idx = torch.arange(y.shape[0], device=x.device) // rb_div
            y = y + rowbias.float()[idx]
        if residual is not None:
            y = y + residual.float()
        return self._o(y), Ho, Wo

    def flash_attn(self, q, k, v, qmap, kmap, groups, heads, q_len, kv_len, *, out=None, out_scale=1.0, accumulate=False, causal=False,
                   accumulation_target=False):
        C = q.shape[1]
        D = C // heads
        qi = rowmap_indices(qmap, groups, q_len).to(q.device)
        ki = rowmap_indices(kmap, groups, kv_len).to(q.device)
        qg = q.float()[qi].reshape(groups, q_len, heads, D).transpose(1, 2)
        kg = k.float()[ki].reshape(groups, kv_len, heads, D).transpose(1, 2)
        vg = v.float()[ki].reshape(groups, kv_len, heads, D).transpose(1, 2)
        sc = qg @ kg.transpose(-1, -2) * (D ** -0.5)
        if causal:
            sc = sc.masked_fill(torch.arange(kv_len, device=q.device)[None, :] > torch.arange(q_len, device=q.device)[:, None], float("-inf"))
        p = torch.softmax(sc, dim=-1)
        o = (p @ vg).transpose(1, 2).reshape(groups, q_len, C) * out_scale
        res = out if out is not None else torch.zeros((q.shape[0], C), dtype=self.act_dtype, device=q.device)
        flat = qi.reshape(-1)
        if accumulate:
            res[flat] = self._o(res[flat].float() + o.reshape(-1, C))
        else:
            res[flat] = self._o(o.reshape(-1, C))
        return res

    def temporal_attn(self, q, k, v, videos, frames, L, heads, *, q_f0=0, q_frames=None, out=None):
        C = q.shape[1]
        D = C // heads
        fq = frames if q_frames is None else q_frames

        def seq(t, f):     # [(v f) l, C] -> [v, l, heads, f, D]
            return t.float().reshape(videos, f, L, heads, D).permute(0, 2, 3, 1, 4)

        if fq != frames:   # k / v: [frames / fq blocks][videos, fq, L] -> (v f) l order
            nb = frames // fq
            k = k.reshape(nb, videos, fq, L, C).permute(1, 0, 2, 3, 4).reshape(videos * frames * L, C)
            v = v.reshape(nb, videos, fq, L, C).permute(1, 0, 2, 3, 4).reshape(videos * frames * L, C)
        p = torch.softmax(seq(q, fq) @ seq(k, frames).transpose(-1, -2) * (D ** -0.5), dim=-1)
        o = (p @ seq(v, frames)).permute(0, 3, 1, 2, 4).reshape(videos * fq * L, C)
        if out is not None:
            out.copy_(self._o(o))
            return out
        return self._o(o)

    def group_norm_sums(self, x, B, rows, groups):
        C = x.shape[1]
        t = x.double().reshape(B, rows, groups, C // groups)
        return torch.stack([t.sum(dim=(1, 3)), (t * t).sum(dim=(1, 3))], dim=-1)

    def group_norm_apply(self, x, B, rows, gamma, beta, groups, stats, silu):
        C = x.shape[1]
        t = x.float().reshape(B, rows, groups, C // groups)
        y = ((t - stats[:, None, :, None, 0]) * stats[:, None, :, None, 1]).reshape(B * rows, C) * gamma.float() + beta.float()
        if silu:
            y = F.silu(y)
        return self._o(y)

    def group_norm(self, x, B, rows, gamma, beta, groups, eps, silu):
        C = x.shape[1]
        t = x.float().reshape(B, rows, groups, C // groups)
        mean = t.mean(dim=(1, 3), keepdim=True)
        var = t.var(dim=(1, 3), unbiased=False, keepdim=True)
        y = ((t - mean) / torch.sqrt(var + eps)).reshape(B * rows, C) * gamma.float() + beta.float()
        if silu:
            y = F.silu(y)
        return self._o(y)

    def layer_norm(self, x, gamma, beta, eps, pe1=None, pe1_div=1, pe2=None, pe2_div=1, two=False):
        y = F.layer_norm(x.float(), (x.shape[1],), gamma.float(), beta.float(), eps)
        m = torch.arange(x.shape[0], device=x.device)
        y1 = y if pe1 is None else y + pe1.float()[(m // pe1_div) % pe1.shape[0]]
        if not two:
            return self._o(y1)
        y2 = y if pe2 is None else y + pe2.float()[(m // pe2_div) % pe2.shape[0]]
        return self._o(y1), self._o(y2)

    def geglu(self, x):
        h, gate = x.float().chunk(2, dim=-1)
        return self._o(h * F.gelu(gate))

    def silu(self, x):
        return self._o(F.silu(x.float()))

    def activation(self, x, kind):
        xf = x.float()
        y = {"silu": F.silu, "quick_gelu": lambda t: t * torch.sigmoid(1.702 * t), "gelu": F.gelu}[kind](xf)
        return self._o(y)

    def concat(self, a, b):
        return torch.cat([a, b], dim=1)

    def timestep_embed(self, t, dim):
        half = dim // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        a = t.float()[:, None] * freqs[None]
        return self._o(torch.cat([torch.cos(a), torch.sin(a)], dim=-1))

    def im2col_in(self, sample):
        V, C, Fr, H, W = sample.shape
        img = sample.float().permute(0, 2, 1, 3, 4).reshape(V * Fr, C, H, W)
        cols = F.unfold(img, kernel_size=3, padding=1)                    # [B, C*9, H*W], index c*9 + tap
        cols = cols.reshape(V * Fr, C, 9, H * W).permute(0, 3, 2, 1).reshape(V * Fr * H * W, 9 * C)   # k = tap*C + c
        out = torch.zeros((cols.shape[0], 64), dtype=torch.float32, device=sample.device)
        out[:, : 9 * C] = cols
        return self._o(out)

    def unpack_out(self, x, V, C, Fr, H, W, dtype):
        return x.float().reshape(V, Fr, H, W, C).permute(0, 4, 1, 2, 3).contiguous().to(dtype)

    def gemm_f32out(self, x, w, bias=None, alpha=1.0):
        y = x.float() @ w.float().t()
        if bias is not None:
            y = y + bias.float()
        return (alpha * y).float()

    def softmax_rows(self, x, out=None):
        y = self._o(torch.softmax(x.float(), dim=-1))
        if out is not None:
            out.copy_(y)
            return out
        return y

    def channel_mix(self, x, w, bias, scale=1.0):
        y = scale * torch.einsum("oc,bchw->bohw", w.float().to(x.device), x.float())
        return y if bias is None else y + bias.float().to(x.device)[None, :, None, None]

    def cfg_ddim_step(self, eps_pair, x, first_frame, guidance, alpha_t, alpha_prev):
        n = x.shape[0]
        eu, et = eps_pair[:n], eps_pair[n:]
        e = eu + guidance * (et - eu)
        x0 = (x - math.sqrt(1 - alpha_t) * e) / math.sqrt(alpha_t)
        y = math.sqrt(alpha_prev) * x0 + math.sqrt(1 - alpha_prev) * e
        y[:, :, 0] = first_frame.reshape(n, x.shape[1], x.shape[3], x.shape[4])
        return y

    # ------------------------------------------------------------------ training path (include/animate3d_hip.h "Training path")
    # References of the backward kernels: torch autograd through the forward references above, in fp32.
    def flash_attn2(self, q, k, v, k2, v2, qmap, kmap, kmap2, groups, heads, q_len, kv_len, kv_len2, *, out_scale2=1.0):
        o = self.flash_attn(q, k, v, qmap, kmap, groups, heads, q_len, kv_len)
        return self.flash_attn(q, k2, v2, qmap, kmap2, groups, heads, q_len, kv_len2, out=o, out_scale=out_scale2, accumulate=True)

    def flash_attn_bwd(self, q, k, v, do, qmap, kmap, groups, heads, q_len, kv_len, *, q_per_kv=1, do_scale=1.0, need_dq=True, need_dkv=True,
                       dq_out=None, dk_out=None, dv_out=None):
        with torch.enable_grad():
            qf, kf, vf = (t.detach().float().clone().requires_grad_(True) for t in (q, k, v))
            o = TorchRefOps(torch.float32, q.device).flash_attn(qf, kf, vf, qmap, kmap, groups, heads, q_len, kv_len, out_scale=do_scale)
            o.backward(do.float())
        z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad

        def put(g, out):          # HipOps writes the caller's [rows, C] view (own row stride) and returns it
            g = self._o(g)
            if out is None:
                return g
            out.copy_(g)
            return out
        assert (dk_out is None) == (dv_out is None)
        return (put(z(qf), dq_out) if need_dq else None, put(z(kf), dk_out) if need_dkv else None, put(z(vf), dv_out) if need_dkv else None)

    def temporal_attn_bwd(self, q, k, v, do, videos, frames, L, heads):
        with torch.enable_grad():
            qf, kf, vf = (t.detach().float().clone().requires_grad_(True) for t in (q, k, v))
            TorchRefOps(torch.float32, q.device).temporal_attn(qf, kf, vf, videos, frames, L, heads).backward(do.float())
        return self._o(torch.cat([qf.grad, kf.grad, vf.grad], dim=1))

    def layer_norm_bwd(self, x, dy, gamma, eps, need_param=True):
        with torch.enable_grad():
            xf = x.detach().float().clone().requires_grad_(True)
            g = gamma.detach().float().clone().requires_grad_(True)
            b = torch.zeros_like(g).requires_grad_(True)
            F.layer_norm(xf, (x.shape[1],), g, b, eps).backward(dy.float())
        return self._o(xf.grad), (g.grad if need_param else None), (b.grad if need_param else None)

    def group_norm_stats(self, x, B, rows, groups, eps):
        t = x.float().reshape(B, rows, groups, x.shape[1] // groups)
        mean = t.mean(dim=(1, 3))
        var = t.var(dim=(1, 3), unbiased=False)
        return torch.stack([mean, 1.0 / torch.sqrt(var + eps)], dim=-1).float().contiguous()

    def group_norm_bwd(self, x, dy, B, rows, gamma, beta, groups, stats, silu, need_param=False):
        C = x.shape[1]
        with torch.enable_grad():
            xf = x.detach().float().clone().requires_grad_(True)
            g = gamma.detach().float().clone().requires_grad_(True)
            b = beta.detach().float().clone().requires_grad_(True)
            t = xf.reshape(B, rows, groups, C // groups)
            mean = t.mean(dim=(1, 3), keepdim=True)
            var = t.var(dim=(1, 3), unbiased=False, keepdim=True)
            eps = (1.0 / stats[:, None, :, None, 1] ** 2 - var).detach()          # the eps the statistics were taken with
            y = ((t - mean) / torch.sqrt(var + eps)).reshape(B * rows, C) * g + b
            if silu:
                y = F.silu(y)
            y.backward(dy.float())
        return self._o(xf.grad), (g.grad if need_param else None), (b.grad if need_param else None)

    def geglu_bwd(self, proj_il, dy):
        N = proj_il.shape[1] // 2
        with torch.enable_grad():
            p = proj_il.detach().float().clone().requires_grad_(True)
            blk = p.reshape(p.shape[0], N // 32, 2, 32)
            (blk[:, :, 0].reshape(-1, N) * F.gelu(blk[:, :, 1].reshape(-1, N))).backward(dy.float())
        return self._o(p.grad)

    def transpose(self, x, pad=64):
        rows, cols = x.shape
        rp = (rows + pad - 1) // pad * pad
        y = torch.zeros((cols, rp), dtype=x.dtype, device=x.device)
        y[:, :rows] = x.t()
        return y

    def wgrad(self, dy, x, alpha=1.0):
        return alpha * (dy.float().t() @ x.float())

    def colsum(self, x, alpha=1.0):
        return alpha * x.float().sum(dim=0)

    def axpby_(self, x, y, a=1.0, b=1.0):
        y.copy_(self._o(a * x.float() + (b * y.float() if b != 0.0 else 0.0)))
        return y

    def scaled(self, x, a):
        return self._o(a * x.float())

    def zero_insert2x(self, dy, B, H, W):
        C = dy.shape[1]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        z = torch.zeros((B, H, W, C), dtype=dy.dtype, device=dy.device)
        z[:, ::2, ::2] = dy.reshape(B, Ho, Wo, C)
        return z.reshape(B * H * W, C)

    def upsample2x_bwd(self, du, B, H, W, He, We):
        C = du.shape[1]
        with torch.enable_grad():
            x = torch.zeros((B, C, H, W), dtype=torch.float32, device=du.device, requires_grad=True)
            F.interpolate(x, size=(He, We), mode="nearest").backward(du.float().reshape(B, He, We, C).permute(0, 3, 1, 2))
        return self._o(x.grad.permute(0, 2, 3, 1).reshape(B * H * W, C))

    def sqnorm(self, g, out=None, accumulate=False):
        s = (g.float() ** 2).sum().reshape(1)
        if out is not None:
            out.copy_(out + s if accumulate else s)
            return out
        return s

    def clip_ctrl(self, sq, max_norm, inv_loss_scale=1.0):
        norm = torch.sqrt(sq[0]) * inv_loss_scale
        finite = bool(torch.isfinite(norm))
        coef = min(1.0, max_norm / (float(norm) + 1e-6)) if (max_norm > 0 and finite) else 1.0
        return torch.tensor([inv_loss_scale * coef if finite else 0.0, 0.0 if finite else 1.0, float(norm)], dtype=torch.float32, device=sq.device)

    def adamw_(self, p, g, m, v, *, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, step=1, ctrl=None):
        if ctrl is not None and float(ctrl[1]) != 0.0:
            return p
        gs = g * (float(ctrl[0]) if ctrl is not None else 1.0)
        b1, b2 = betas
        p.mul_(1.0 - lr * weight_decay)
        m.mul_(b1).add_(gs, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(gs, gs, value=1.0 - b2)
        denom = v.sqrt() / math.sqrt(1.0 - b2 ** step) + eps
        p.addcdiv_(m, denom, value=-lr / (1.0 - b1 ** step))
        return p
