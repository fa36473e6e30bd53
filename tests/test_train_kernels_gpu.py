"""GPU parity tests of the training path's kernels (include/animate3d_hip.h "Training path"): every backward entry
point against torch autograd through the plain-PyTorch fp32 reference of the forward op (tests/torch_ops.py) on the SAME 16-bit
inputs, through the C-ABI, for both storage types; then the autograd compositions of animate3d_amd/autograd_ops.py over the real
kernels against the same class over the reference op set.

Tolerances: the gradients are rounded to the storage type once on the way out (2^-9 bf16 / 2^-12 fp16 relative) and P / dS are
rounded to it in front of the gradient products (as in every flash-attention backward); reference in fp32.  Bars: relative L2
2e-2 (bf16) / 3e-3 (fp16) for attention gradients, 6e-3 / 1e-3 for the normalisation and elementwise kernels.
"""
import math

import pytest
import torch

from animate3d_amd.hip_ops import RowMap
from tests.torch_ops import TorchRefOps

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]
ATTN_TOL = {torch.bfloat16: 2e-2, torch.float16: 3e-3}
ELEM_TOL = {torch.bfloat16: 6e-3, torch.float16: 1e-3}


@pytest.fixture(scope="module", params=DTYPES, ids=["bf16", "fp16"])
def ops(request):
    from animate3d_amd.hip_ops import HipOps
    return HipOps(act_dtype=request.param)


@pytest.fixture(scope="module")
def ref():
    return TorchRefOps(act_dtype=torch.float32, device="cuda")


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, generator=g, device="cuda", dtype=torch.float32) * scale).to(dtype)


def check(name, got, want, tol):
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert torch.isfinite(got.float()).all(), f"{name}: non-finite output"
    e = ((got.float() - want.float()).norm() / (want.float().norm() + 1e-20)).item()
    mx = (got.float() - want.float()).abs().max().item()
    print(f"[parity] {name}: rel_l2={e:.3e} max_abs={mx:.3e} (ref max {want.float().abs().max().item():.3e})")
    assert e <= tol, f"{name}: rel L2 error {e:.3e} > {tol:.1e}"


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("C,n,F,L", [(320, 2, 2, 96), (640, 2, 3, 40), (1280, 3, 2, 16), (320, 2, 2, 328), (640, 2, 2, 272), (320, 2, 2, 128), (1280, 2, 2, 64)])   # ragged tails; last two: tiles aligned to the map segments (no per-row division)
def test_flash_attn_bwd_multiview_and_first_frame(ops, ref, C, n, F, L):
    """attention_processor.py:340, 389-418: "(b n f) l -> (b f) (n l)" queries; the first-frame branch reads frame 0's keys for all
    F frames of a video, so dK / dV are sums over F query groups and vanish on the other frames."""
    dt, b, heads = ops.act_dtype, 2, 8
    rows = b * n * F * L
    kvq = rnd(rows, 3 * C, seed=1, dtype=dt)
    q, k, v = kvq[:, 2 * C:], kvq[:, :C], kvq[:, C:2 * C]
    do = rnd(rows, C, seed=2, dtype=dt)
    qm = RowMap(gdiv=F, ga=n * F * L, gb=L, seg_len=L, seg_stride=F * L)
    k0 = RowMap(gdiv=F, ga=n * F * L, gb=0, seg_len=L, seg_stride=F * L)
    for name, km, share in (("multi-view", qm, 1), ("first-frame", k0, F)):
        got = ops.flash_attn_bwd(q, k, v, do, qm, km, b * F, heads, n * L, n * L, q_per_kv=share)
        want = ref.flash_attn_bwd(q, k, v, do, qm, km, b * F, heads, n * L, n * L)
        for g, w, t in zip(got, want, "qkv"):
            check(f"flash_attn_bwd {name} d{t} D={C // heads}", g, w, ATTN_TOL[dt])
    fr = torch.arange(rows, device="cuda") // L % F
    assert float(got[1][fr != 0].abs().max()) == 0.0 and float(got[2][fr != 0].abs().max()) == 0.0


def _lse_reference(q, k, qmap, kmap, groups, heads, q_len, kv_len):
    """log2 sum_k 2^(q.k * scale * log2 e) per (group, head, query), fp32, through the row maps."""
    C = q.shape[1]
    D = C // heads
    out = torch.empty(groups, heads, q_len, device=q.device)
    for g in range(groups):
        rows = lambda m, n: (g // m.gdiv) * m.ga + (g % m.gdiv) * m.gb + (torch.arange(n, device=q.device) // m.seg_len) * m.seg_stride + torch.arange(n, device=q.device) % m.seg_len
        qr, kr = rows(qmap, q_len), rows(kmap, kv_len)
        qh = q[qr].float().reshape(q_len, heads, D).permute(1, 0, 2)
        kh = k[kr].float().reshape(kv_len, heads, D).permute(1, 0, 2)
        s = qh @ kh.transpose(1, 2) * (D ** -0.5)
        out[g] = torch.logsumexp(s, dim=-1) / math.log(2.0)
    return out


@pytest.mark.parametrize("C,n,F,L", [(320, 4, 2, 256), (320, 2, 2, 96), (640, 4, 2, 256), (640, 2, 3, 40), (1280, 3, 2, 16), (1280, 2, 2, 64), (1280, 4, 2, 64)])
def test_flash_attn_log_sum_exp_feeds_the_backward(ops, ref, C, n, F, L):
    """a3d_flash_attn_lse (every kernel that serves the training shapes: the LDS-DMA kernels at head_dim 40 / 80 / 160 from 256 / 512 / 256 keys,
    the plain kernel elsewhere) returns the log2 softmax denominator per query; with it and a3d_attn_delta (rowsum(dO * O) per head)
    a3d_flash_attn_bwd skips its statistics pass: same gradients as the recomputing path, multi-view and first-frame maps."""
    dt, b, heads = ops.act_dtype, 2, 8
    rows = b * n * F * L
    kvq = rnd(rows, 3 * C, seed=1, dtype=dt)
    q, k, v = kvq[:, 2 * C:], kvq[:, :C], kvq[:, C:2 * C]
    do = rnd(rows, C, seed=2, dtype=dt)
    qm = RowMap(gdiv=F, ga=n * F * L, gb=L, seg_len=L, seg_stride=F * L)
    k0 = RowMap(gdiv=F, ga=n * F * L, gb=0, seg_len=L, seg_stride=F * L)
    S, G = n * L, b * F
    for name, km, share in (("multi-view", qm, 1), ("first-frame", k0, F)):
        o_plain = ops.flash_attn(q, k, v, qm, km, G, heads, S, S)
        o, lse = ops.flash_attn(q, k, v, qm, km, G, heads, S, S, with_lse=True)
        assert torch.equal(o, o_plain)
        want_lse = _lse_reference(q, k, qm, km, G, heads, S, S)
        err = float((lse - want_lse).abs().max())
        print(f"[parity] lse {name} D={C // heads} S={S}: max abs {err:.3e} log2 units")
        assert err <= (2e-2 if dt == torch.bfloat16 else 4e-3)          # the row sum adds P rounded to 16 bits; Q is pre-scaled in 16 bits at head_dim 40
        slow = ops.flash_attn_bwd(q, k, v, do, qm, km, G, heads, S, S, q_per_kv=share)
        fast = ops.flash_attn_bwd(q, k, v, do, qm, km, G, heads, S, S, q_per_kv=share, o=o, lse=lse)
        want = ref.flash_attn_bwd(q, k, v, do, qm, km, G, heads, S, S)
        for a, bb, w, t in zip(fast, slow, want, "qkv"):
            check(f"flash_attn_bwd with forward statistics {name} d{t} D={C // heads}", a, w, ATTN_TOL[dt])
            check(f"  ... against the recomputing path d{t}", a, bb, ATTN_TOL[dt])
    # out_scale: O carries it, delta = rowsum(dO * O) needs no correction
    o, lse = ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, out_scale=0.6, with_lse=True)
    fast = ops.flash_attn_bwd(q, k, v, do, qm, qm, G, heads, S, S, do_scale=0.6, o=o, lse=lse)
    want = ref.flash_attn_bwd(q, k, v, do, qm, qm, G, heads, S, S, do_scale=0.6)
    for a, w, t in zip(fast, want, "qkv"):
        check(f"flash_attn_bwd with forward statistics, out_scale 0.6, d{t}", a, w, ATTN_TOL[dt])


@pytest.mark.parametrize("C,T,L", [(320, 77, 50), (640, 16, 50), (1280, 77, 50), (320, 77, 600)])
def test_flash_attn_bwd_cross_attention_query_only(ops, ref, C, T, L):
    """attention_processor.py:233-270: text / IP tokens are frozen inputs, only dQ is wanted; the IP branch's out_scale scales dO."""
    dt, heads, V, F = ops.act_dtype, 8, 2, 3
    B2 = V * F
    q, do = rnd(B2 * L, C, seed=1, dtype=dt), rnd(B2 * L, C, seed=2, dtype=dt)
    kv = rnd(V * T, 2 * C, seed=3, dtype=dt)
    qc, kc = RowMap(gdiv=1, ga=L, gb=0, seg_len=L, seg_stride=0), RowMap(F, T, 0, T, 0)
    got = ops.flash_attn_bwd(q, kv[:, :C], kv[:, C:], do, qc, kc, B2, heads, L, T, q_per_kv=F, do_scale=0.6, need_dkv=False)
    want = ref.flash_attn_bwd(q, kv[:, :C], kv[:, C:], do, qc, kc, B2, heads, L, T, do_scale=0.6)
    assert got[1] is None and got[2] is None
    check(f"flash_attn_bwd cross dq D={C // heads} T={T}", got[0], want[0], ATTN_TOL[dt])
    full = ops.flash_attn_bwd(q, kv[:, :C], kv[:, C:], do, qc, kc, B2, heads, L, T, q_per_kv=F, do_scale=0.6)
    check("flash_attn_bwd cross dk", full[1], want[1], ATTN_TOL[dt])
    check("flash_attn_bwd cross dv", full[2], want[2], ATTN_TOL[dt])


@pytest.mark.parametrize("C,F,L", [(320, 16, 9), (640, 5, 7), (1280, 16, 4), (320, 24, 3)])
def test_temporal_attn_bwd(ops, ref, C, F, L):
    dt, videos, heads = ops.act_dtype, 2, 8
    qkv = rnd(videos * F * L, 3 * C, seed=1, dtype=dt)
    do = rnd(videos * F * L, C, seed=2, dtype=dt)
    got = ops.temporal_attn_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], do, videos, F, L, heads)
    want = ref.temporal_attn_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], do, videos, F, L, heads)
    for i, t in enumerate("qkv"):
        check(f"temporal_attn_bwd d{t} C={C} F={F}", got[:, i * C:(i + 1) * C], want[:, i * C:(i + 1) * C], ELEM_TOL[dt] * 2)


# ------------------------------------------------------------------ normalisation / elementwise
@pytest.mark.parametrize("M,C", [(100, 320), (37, 640), (260, 1280), (16, 768)])
def test_layer_norm_bwd(ops, ref, M, C):
    dt = ops.act_dtype
    x, dy = rnd(M, C, seed=1, scale=2.0, dtype=dt), rnd(M, C, seed=2, dtype=dt)
    gamma = 1.0 + 0.2 * rnd(C, seed=3)
    dx, dg, db = ops.layer_norm_bwd(x, dy, gamma, 1e-5)
    rx, rg, rb = ref.layer_norm_bwd(x, dy, gamma, 1e-5)
    check(f"layer_norm_bwd dx {M}x{C}", dx, rx, ELEM_TOL[dt])
    check("layer_norm_bwd dgamma", dg, rg, 1e-4)
    check("layer_norm_bwd dbeta", db, rb, 1e-4)
    dx2, dg2, _ = ops.layer_norm_bwd(x, dy, gamma, 1e-5, need_param=False)            # frozen affine: its own kernel instantiation
    assert dg2 is None
    check(f"layer_norm_bwd dx (no parameter gradients) {M}x{C}", dx2, rx, ELEM_TOL[dt])


@pytest.mark.parametrize("M,C", [(4099, 320), (1030, 640), (515, 1280)])
def test_layer_norm_bwd_sub_wave_rows_kernel_tail_and_grid_stride(ops, ref, M, C):
    """C = 40 * LPR: LPR lanes per row, 64 / LPR rows per wave step; M is not a multiple of the rows per step (masked tail rows must not
    reach dgamma / dbeta) and, with parameter gradients, large enough that the <= 512 workgroups loop."""
    dt = ops.act_dtype
    x, dy = rnd(M, C, seed=11, scale=1.5, dtype=dt) + 0.25, rnd(M, C, seed=12, dtype=dt)
    gamma = 1.0 + 0.2 * rnd(C, seed=13)
    rx, rg, rb = ref.layer_norm_bwd(x, dy, gamma, 1e-5)
    for need in (True, False):
        dx, dg, db = ops.layer_norm_bwd(x, dy, gamma, 1e-5, need_param=need)
        check(f"layer_norm_bwd dx {M}x{C} need_param={need}", dx, rx, ELEM_TOL[dt])
        if need:
            check("layer_norm_bwd dgamma", dg, rg, 1e-4)
            check("layer_norm_bwd dbeta", db, rb, 1e-4)


@pytest.mark.parametrize("rows,cols,pad,off", [(1280, 5120, 1, 0), (2560, 320, 1, 0), (328, 320, 1, 8), (100, 72, 64, 0), (1000, 1288, 8, 16), (8, 8, 1, 0)])
def test_transpose_16_byte_kernel(ops, ref, rows, cols, pad, off):
    """W -> W^T of the trainable weights (every step) and the padded activations of the weight-gradient GEMM: 16-byte loads and stores when
    cols, the padded row count and both leading dimensions are multiples of 8; bit-exact, zero padding included."""
    xb = rnd(rows, cols + off, seed=31, dtype=ops.act_dtype)
    x = xb[:, off:] if off else xb
    got, want = ops.transpose(x, pad=pad), ref.transpose(x, pad=pad)
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize("M,N,pad", [(1000, 320, 0), (777, 640, 24), (300, 2560, 0), (129, 3840, 8), (50, 960, 0), (70000, 1280, 0), (3, 320, 0)])
def test_colsum_16_byte_kernel(ops, ref, M, N, pad):
    """Bias gradients of widths that are multiples of 320 (every Linear of the model): 16-byte loads, TPR lanes per row."""
    dt = ops.act_dtype
    xb = rnd(M, N + pad, seed=21, dtype=dt)
    x = xb[:, pad:] if pad else xb
    check(f"colsum {M}x{N} ld={x.stride(0)}", ops.colsum(x, 0.5), ref.colsum(x, 0.5), 1e-4 if M > 10000 else 1e-5)


@pytest.mark.parametrize("B,rows,C,silu", [(2, 64, 320, True), (3, 50, 960, True), (2, 4 * 16, 640, False), (1, 33, 2560, True)])
def test_group_norm_bwd(ops, ref, B, rows, C, silu):
    """ResnetBlock2D.norm1/2 (+SiLU), Transformer2DModel.norm and the per-video 3-D norm of the motion modules (rows = F * h * w)."""
    dt, groups = ops.act_dtype, 32
    x = (rnd(B * rows, C, seed=1, scale=1.5) + 0.3).to(dt)
    dy = rnd(B * rows, C, seed=2, dtype=dt)
    gamma, beta = 1.0 + 0.2 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    stats = ops.group_norm_stats(x, B, rows, groups, 1e-5)
    check("group_norm_stats", stats, ref.group_norm_stats(x, B, rows, groups, 1e-5), 1e-5)
    dx, dg, db = ops.group_norm_bwd(x, dy, B, rows, gamma, beta, groups, stats, silu, need_param=True)
    rx, rg, rb = ref.group_norm_bwd(x, dy, B, rows, gamma, beta, groups, stats, silu, need_param=True)
    check(f"group_norm_bwd dx B={B} rows={rows} C={C} silu={silu}", dx, rx, ELEM_TOL[dt])
    check("group_norm_bwd dgamma", dg, rg, 1e-4)
    check("group_norm_bwd dbeta", db, rb, 1e-4)


def test_geglu_bwd_and_layout_helpers(ops, ref):
    dt = ops.act_dtype
    p, dy = rnd(70, 2560, seed=1, dtype=dt), rnd(70, 1280, seed=2, dtype=dt)
    check("geglu_bwd", ops.geglu_bwd(p, dy), ref.geglu_bwd(p, dy), ELEM_TOL[dt])
    x = rnd(100, 328, seed=3, dtype=dt)
    t = ops.transpose(x[:, 8:])                                  # strided input, rows padded 100 -> 128 with zeros
    assert t.shape == (320, 128) and torch.equal(t, ref.transpose(x[:, 8:]))
    assert torch.equal(ops.transpose(x, pad=1), x.t().contiguous())
    check("colsum", ops.colsum(x, 0.5), ref.colsum(x, 0.5), 1e-5)
    big = rnd(5000, 72, seed=4, dtype=dt)
    check("colsum tall", ops.colsum(big), ref.colsum(big), 1e-5)
    y = rnd(100, 328, seed=5, dtype=dt)
    check("axpby", ops.axpby_(x, y.clone(), 0.5, 2.0), 0.5 * x.float() + 2.0 * y.float(), ELEM_TOL[dt])
    check("scaled", ops.scaled(x, -0.25), -0.25 * x.float(), ELEM_TOL[dt])
    for H, W in ((6, 6), (5, 7)):
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        d = rnd(2 * Ho * Wo, 64, seed=6, dtype=dt)
        assert torch.equal(ops.zero_insert2x(d, 2, H, W), ref.zero_insert2x(d, 2, H, W))
    for He, We in ((6, 10), (5, 9), (6, 9)):
        du = rnd(2 * He * We, 64, seed=7, dtype=dt)
        check(f"upsample2x_bwd {(He, We)}", ops.upsample2x_bwd(du, 2, 3, 5, He, We), ref.upsample2x_bwd(du, 2, 3, 5, He, We), ELEM_TOL[dt])


@pytest.mark.parametrize("M,N,K", [(5000, 320, 320), (300, 2560, 320), (65536, 320, 1280), (77, 8, 72), (4096, 1280, 5120)])
def test_wgrad_split_over_tokens(ops, ref, M, N, K):
    """dW = dY^T X of the trainable Linear layers: natural layouts in (strided column views allowed), fp32 out, token axis split."""
    dt = ops.act_dtype
    dyb, xb = rnd(M, N + 8, seed=1, dtype=dt), rnd(M, K + 16, seed=2, dtype=dt)
    dy, x = dyb[:, 8:], xb[:, :K]
    got, want = ops.wgrad(dy, x, 0.5), ref.wgrad(dy, x, 0.5)
    assert got.dtype == torch.float32
    check(f"wgrad {M}x{N}x{K}", got, want, 2e-5)
    assert torch.equal(ops.wgrad(dy, x, 0.5), got)          # counted vmcnt waits: deterministic run to run


def test_optimizer_kernels_match_torch_adamw(ops):
    """train.py:351-357, 583-596: AdamW over one flat fp32 buffer, clip_grad_norm_ and the GradScaler's skip on non-finite gradients."""
    if ops.act_dtype != torch.bfloat16:
        pytest.skip("dtype-free kernels: tested once")
    n = 100_003
    p0 = rnd(n, seed=1)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = rnd(n, seed=10 + step, scale=3.0)
        ref_p.grad = g.clone()
        total = torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        ctrl = ops.clip_ctrl(ops.sqnorm(g), 1.0, 1.0)
        assert abs(float(ctrl[2]) - float(total)) < 1e-4 * float(total) and float(ctrl[1]) == 0.0
        ops.adamw_(p, g, m, v, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, step=step, ctrl=ctrl)
        err = (p - ref_p.detach()).abs().max().item()
        print(f"[parity] adamw step {step}: max |p - torch| = {err:.2e}")
        assert err < 2e-6
    # loss-scaled gradients: unscale inside the step; a non-finite gradient skips it
    g = rnd(n, seed=20) * 1024.0
    ctrl = ops.clip_ctrl(ops.sqnorm(g), 0.0, 1.0 / 1024.0)
    assert abs(float(ctrl[0]) - 1.0 / 1024.0) < 1e-9
    g[5] = float("inf")
    before = p.clone()
    ctrl = ops.clip_ctrl(ops.sqnorm(g), 1.0, 1.0 / 1024.0)
    ops.adamw_(p, g, m, v, lr=1e-3, step=4, ctrl=ctrl)
    assert float(ctrl[1]) == 1.0 and torch.equal(p, before)


# ------------------------------------------------------------------ the autograd compositions over the real kernels
def _grads(fn, tensors, dy):
    leaves = [t.detach().clone().requires_grad_(True) for t in tensors]
    out = fn(*leaves)
    out.backward(dy.to(out.dtype))
    return [t.grad for t in leaves]


def test_autograd_ops_gemm_conv_compositions(ops, ref):
    from animate3d_amd.autograd_ops import AutogradOps
    dt = ops.act_dtype
    a, r = AutogradOps(ops), AutogradOps(ref)
    tol = ATTN_TOL[dt]
    # gemm with bias, residual and a trainable merge weight
    x, w, res = rnd(192, 320, seed=1, dtype=dt), rnd(640, 320, seed=2, scale=320 ** -0.5, dtype=dt), rnd(192, 640, seed=3, dtype=dt)
    bias, alpha, dy = 0.1 * rnd(640, seed=4), torch.tensor(0.6, device="cuda"), rnd(192, 640, seed=5)
    got = _grads(lambda x, w, b, res, al: a.gemm(x, w, b, residual=res, alpha=al), [x, w, bias, res, alpha], dy)
    want = _grads(lambda x, w, b, res, al: r.gemm(x, w, b, residual=res, alpha=al), [x.float(), w.float(), bias, res.float(), alpha], dy)
    for name, g, wnt in zip(("dx", "dw", "db", "dres", "dalpha"), got, want):
        check(f"autograd gemm {name}", g.reshape(-1), wnt.reshape(-1), tol if name != "dalpha" else 2e-2)
    # gemm_geglu (interleaved weights)
    wg, bg = rnd(2560, 320, seed=6, scale=320 ** -0.5, dtype=dt), 0.1 * rnd(2560, seed=7)
    dy = rnd(192, 1280, seed=8)
    got = _grads(lambda x, w, b: a.gemm_geglu(x, w, b), [x, wg, bg], dy)
    want = _grads(lambda x, w, b: r.gemm_geglu(x, w, b), [x.float(), wg.float(), bg], dy)
    for name, g, wnt in zip(("dx", "dw", "db"), got, want):
        check(f"autograd gemm_geglu {name}", g, wnt, tol)
    # conv3x3 input gradients: stride 1 (+ residual), stride 2 on an odd size, forced up-sample size, conv_out (4 channels)
    B, H, W = 2, 7, 6
    xc = rnd(B * H * W, 64, seed=9, dtype=dt)
    wc = rnd(128, 9 * 64, seed=10, scale=(9 * 64) ** -0.5, dtype=dt)
    cases = [("stride1", dict(), (H, W)), ("stride2", dict(stride=2), ((H - 1) // 2 + 1, (W - 1) // 2 + 1)),
             ("up2x", dict(up2x=True), (2 * H, 2 * W)), ("up forced", dict(up2x=True, up_size=(2 * H - 1, 2 * W)), (2 * H - 1, 2 * W))]
    for name, kw, (Ho, Wo) in cases:
        dy = rnd(B * Ho * Wo, 128, seed=11)
        res = rnd(B * Ho * Wo, 128, seed=12, dtype=dt)
        got = _grads(lambda x, res: a.conv3x3(x, B, H, W, wc, None, residual=res, **kw)[0], [xc, res], dy)
        want = _grads(lambda x, res: r.conv3x3(x, B, H, W, wc.float(), None, residual=res, **kw)[0], [xc.float(), res.float()], dy)
        check(f"autograd conv3x3 {name} dx", got[0], want[0], tol)
        check(f"autograd conv3x3 {name} dres", got[1], want[1], tol)
    w4 = rnd(4, 9 * 64, seed=13, scale=(9 * 64) ** -0.5, dtype=dt)
    dy = rnd(B * H * W, 4, seed=14)
    got = _grads(lambda x: a.conv3x3(x, B, H, W, w4, None)[0], [xc], dy)
    want = _grads(lambda x: r.conv3x3(x, B, H, W, w4.float(), None)[0], [xc.float()], dy)
    check("autograd conv3x3 conv_out dx", got[0], want[0], tol)


def test_autograd_ops_attention_and_norm_compositions(ops, ref):
    """The fused-projection views unet.py hands to the attention ops, the IP-adapter accumulation into one buffer and the
    two-output LayerNorm, differentiated through the real kernels."""
    from animate3d_amd.autograd_ops import AutogradOps
    dt = ops.act_dtype
    a, r = AutogradOps(ops), AutogradOps(ref)
    tol = ATTN_TOL[dt]
    C, heads, b, n, F, L = 320, 8, 1, 2, 2, 32
    V = b * n
    rows = V * F * L
    qm = RowMap(gdiv=F, ga=n * F * L, gb=L, seg_len=L, seg_stride=F * L)
    k0 = RowMap(gdiv=F, ga=n * F * L, gb=0, seg_len=L, seg_stride=F * L)

    def self_attn(o, kvq):       # multi-view + first-frame attention on one [rows, 4C] projection, then the temporal one on [rows, 3C]
        k, v, q, qi = kvq[:, :C], kvq[:, C:2 * C], kvq[:, 2 * C:3 * C], kvq[:, 3 * C:]
        y = o.flash_attn(q, k, v, qm, qm, b * F, heads, n * L, n * L)
        y2 = o.flash_attn(qi, k, v, qm, k0, b * F, heads, n * L, n * L)
        t = o.temporal_attn(kvq[:, :C], kvq[:, C:2 * C], kvq[:, 2 * C:3 * C], V, F, L, heads)
        return torch.cat([y, y2, t], dim=1)

    kvq = rnd(rows, 4 * C, seed=1, dtype=dt)
    dy = rnd(rows, 3 * C, seed=2)
    got = _grads(lambda t: self_attn(a, t), [kvq], dy)
    want = _grads(lambda t: self_attn(r, t), [kvq.float()], dy)
    check("autograd self-attention d(kvq)", got[0], want[0], tol)

    T, nt = 77, 4
    qc = RowMap(gdiv=1, ga=L, gb=0, seg_len=L, seg_stride=0)

    def cross(o, q2, kvt, kvi):
        ca = o.flash_attn(q2, kvt[:, :C], kvt[:, C:], qc, RowMap(F, T, 0, T, 0), V * F, heads, L, T, accumulation_target=True)
        o.flash_attn(q2, kvi[:, :C], kvi[:, C:], qc, RowMap(F, nt, 0, nt, 0), V * F, heads, L, nt, out=ca, out_scale=0.7, accumulate=True)
        return ca

    q2, kvt, kvi = rnd(rows, C, seed=3, dtype=dt), rnd(V * T, 2 * C, seed=4, dtype=dt), rnd(V * nt, 2 * C, seed=5, dtype=dt)
    dy = rnd(rows, C, seed=6)
    got = _grads(lambda q: cross(a, q, kvt, kvi), [q2], dy)
    with pytest.raises(RuntimeError, match="accumulation_target"):      # without the hint the first call keeps its output for the backward: refused loudly
        first = a.flash_attn(q2.clone().requires_grad_(True), kvt[:, :C], kvt[:, C:], qc, RowMap(F, T, 0, T, 0), V * F, heads, L, T)
        a.flash_attn(q2, kvi[:, :C], kvi[:, C:], qc, RowMap(F, nt, 0, nt, 0), V * F, heads, L, nt, out=first, out_scale=0.7, accumulate=True)
    want = _grads(lambda q: cross(r, q, kvt.float(), kvi.float()), [q2.float()], dy)
    check("autograd text + IP cross-attention dq", got[0], want[0], tol)

    x = rnd(rows, C, seed=7, scale=2.0, dtype=dt)
    gamma, beta = 1.0 + 0.2 * rnd(C, seed=8), 0.1 * rnd(C, seed=9)
    pe1, pe2 = rnd(F, C, seed=10, dtype=dt), rnd(L, C, seed=11, dtype=dt)

    def ln2(o, x, g, bt, p1, p2):
        y1, y2 = o.layer_norm(x, g, bt, 1e-5, pe1=p1, pe1_div=L, pe2=p2, pe2_div=1, two=True)
        return torch.cat([y1, 2.0 * y2], dim=1)

    dy = rnd(rows, 2 * C, seed=12)
    got = _grads(lambda x, g, bt: ln2(a, x, g, bt, pe1, pe2), [x, gamma, beta], dy)
    want = _grads(lambda x, g, bt: ln2(r, x, g, bt, pe1.float(), pe2.float()), [x.float(), gamma, beta], dy)
    for name, g, wnt in zip(("dx", "dgamma", "dbeta"), got, want):
        check(f"autograd two-output layer_norm {name}", g, wnt, tol)
    gx = rnd(V * F * L, C, seed=13, scale=1.5, dtype=dt)
    dy = rnd(V * F * L, C, seed=14)
    got = _grads(lambda x, g, bt: a.group_norm(x, V, F * L, g, bt, 32, 1e-6, False), [gx, gamma, beta], dy)
    want = _grads(lambda x, g, bt: r.group_norm(x, V, F * L, g, bt, 32, 1e-6, False), [gx.float(), gamma, beta], dy)
    for name, g, wnt in zip(("dx", "dgamma", "dbeta"), got, want):
        check(f"autograd 3-D group_norm {name}", g, wnt, tol)
