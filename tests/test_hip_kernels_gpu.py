"""GPU parity tests: every entry point of libanimate3d_hip.so against the plain-PyTorch fp32 reference
of the same op (tests/torch_ops.py) on seeded inputs, through the C-ABI (animate3d_amd/hip_ops.py).

Tolerances (bf16 storage, fp32 accumulate): the reference is computed in fp32 from the SAME bf16
inputs, so the only differences are fp32 summation order, the bf16 rounding of P in P·V and the final
bf16 rounding of the output (relative 2^-9 = 2.0e-3).  Bars: relative L2 error <= 4e-3 for GEMM-like
ops and attention, max-abs error <= 2 bf16 ulps of the output scale for normalisation/elementwise ops.
"""

import pytest
import torch

from animate3d_amd.hip_ops import RowMap
from tests.torch_ops import TorchRefOps

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from animate3d_amd.hip_ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def ref():
    return TorchRefOps(act_dtype=torch.float32, device="cuda")


def rnd(*shape, seed=0, scale=1.0, dtype=BF):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, generator=g, device="cuda", dtype=torch.float32) * scale).to(dtype)


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def check(name, got, want, tol=4e-3, max_ulps=None):
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert torch.isfinite(got.float()).all(), f"{name}: non-finite output"
    e = rel_l2(got, want)
    mx = (got.float() - want.float()).abs().max().item()
    scale = want.float().abs().max().item()
    print(f"[parity] {name}: rel_l2={e:.3e} max_abs={mx:.3e} (ref max {scale:.3e})")
    assert e <= tol, f"{name}: rel L2 error {e:.3e} > {tol:.1e} (max abs {mx:.3e})"
    if max_ulps is not None:
        assert mx <= max_ulps * scale * 2.0 ** -8, f"{name}: max abs {mx:.3e} > {max_ulps} bf16 ulps of {scale:.3e}"


# ------------------------------------------------------------------ GEMM / conv
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (389, 1280, 640), (8, 1280, 320), (1000, 4, 320), (128, 128, 64), (130, 2560, 1280)])
def test_gemm_plain(ops, ref, M, N, K):
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias = rnd(N, seed=3, dtype=torch.float32)
    check(f"gemm {M}x{N}x{K}", ops.gemm(x, w, bias), ref.gemm(x, w, bias))
    check(f"gemm nobias {M}x{N}x{K}", ops.gemm(x, w), ref.gemm(x, w))


def test_gemm_epilogues_and_strides(ops, ref):
    M, N, K = 777, 640, 320
    big = rnd(M, 3 * K, seed=4)
    x = big[:, K:2 * K]                       # strided A (a slice of a fused QKV buffer)
    w, bias = rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6, dtype=torch.float32)
    res = rnd(M, N, seed=7)
    check("gemm residual alpha/beta", ops.gemm(x, w, bias, residual=res, alpha=0.37, beta=1.0),
          ref.gemm(x, w, bias, residual=res, alpha=0.37, beta=1.0))
    rb = rnd(7, N, seed=8)
    check("gemm rowbias", ops.gemm(x, w, bias, rowbias=rb, rb_div=111), ref.gemm(x, w, bias, rowbias=rb, rb_div=111))
    out = torch.zeros(M, 2 * N, device="cuda", dtype=BF)
    ops.gemm(x, w, bias, out=out[:, N:])
    check("gemm strided out", out[:, N:], ref.gemm(x, w, bias))
    assert (out[:, :N] == 0).all()


@pytest.mark.parametrize("M,N2,K", [(300, 2560, 320), (1000, 5120, 640), (64, 10240, 1280), (5, 64, 64)])
def test_gemm_geglu_fused(ops, ref, M, N2, K):
    x, w = rnd(M, K, seed=1), rnd(N2, K, seed=2, scale=K ** -0.5)
    bias = rnd(N2, seed=3, dtype=torch.float32)
    w_il, b_il = ops.interleave_geglu(w), ops.interleave_geglu(bias)
    want = ref.geglu(ref.gemm(x, w, bias))                    # un-fused fp32 reference with the original row order
    check(f"gemm_geglu {M}x{N2}x{K}", ops.gemm_geglu(x, w_il, b_il), want)
    check(f"gemm_geglu ref-op {M}x{N2}x{K}", ref.gemm_geglu(x, w_il, b_il), want, tol=1e-6)


def test_gemm_layout_asymmetric(ops):
    """A = I (padded) with an asymmetric W catches operand / output transposes."""
    K = N = 64
    x = torch.eye(K, device="cuda", dtype=BF)
    w = (torch.arange(N * K, device="cuda", dtype=torch.float32).reshape(N, K) % 251 - 125).to(BF)
    y = ops.gemm(x, w)
    assert torch.equal(y, w.t().contiguous())


# The big token matrices (M % 256 == 0, >= 50 % average fill of the persistent grid's rounds) take the persistent LDS-DMA kernel with the
# ping-pong main loop (csrc/gemm_pp.hip); ``tile128=True`` (A3D_GEMM_TILE128 in the call's flags word) forces the 128 x 128-tile kernel, whose
# K order and epilogue arithmetic are identical => bit-equal outputs.
def _persistent_eligible(M, N, geglu=False, cus=256):
    nb = (4 if N % 256 == 0 else 0) if geglu else (5 if N % 320 == 0 else (4 if N % 256 == 0 else 0))
    if nb == 0 or M % 256:
        return False
    if nb == 5 and N % 256 == 0:         # both tile widths divide N: the narrower one when it needs fewer (rounds x width)
        c5, c4 = -(-((M // 256) * (N // 320)) // cus) * 5, -(-((M // 256) * (N // 256)) // cus) * 4
        nb = 4 if c4 * 108 < c5 * 100 else 5
    tiles = (M // 256) * (N // (64 * nb))
    rounds = -(-tiles // cus)
    return tiles * 100 >= rounds * cus * 50           # average fill of the persistent grid's rounds >= 50 %


def _first(t):
    return t[0] if isinstance(t, tuple) else t


def _both_paths(ops, fn):
    """fn(tile128) -> output.  Returns (default dispatch, 128 x 128-tile kernel, default dispatch with 16 CUs reserved).  Split-K is off: it
    re-associates the K sum (its own tests: test_split_k_*)."""
    split, ops.split_k = ops.split_k, False
    try:
        classic = fn(True)
        got = fn(False)
        for _ in range(2):                              # the counted-wait pipeline must be deterministic run to run
            assert torch.equal(_first(fn(False)), _first(got))
        try:
            ops.reserved_cus = 16                       # a smaller persistent grid walks the same tiles
            reserved = fn(False)
        finally:
            ops.reserved_cus = 0
    finally:
        ops.split_k = split
    return got, classic, reserved


@pytest.mark.parametrize("M,N,K", [(65536, 320, 64), (65536, 320, 320), (65536, 640, 192), (65536, 256, 128), (49152, 1280, 128), (24576, 2560, 64),
                                   (8192, 1280, 128)])          # last: one round only -> the 256 x 256 tile (160 tiles) instead of 256 x 320 (128 tiles)
def test_gemm_persistent_path(ops, ref, M, N, K):
    assert _persistent_eligible(M, N)          # same rule as try_launch_persist() in csrc/gemm_conv.hip
    x, w = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=K ** -0.5)
    bias = rnd(N, seed=13, dtype=torch.float32)
    res = rnd(M, N, seed=14)
    rb = rnd(M // 4096, N, seed=15)
    for name, kw in (("plain", {}), ("residual", dict(residual=res, alpha=0.37, beta=1.0)), ("residual beta", dict(residual=res, alpha=0.63, beta=0.9)),
                     ("rowbias+res", dict(rowbias=rb, rb_div=4096, residual=res))):
        got, classic, pinned = _both_paths(ops, lambda t128: ops.gemm(x, w, bias, tile128=t128, **kw))
        check(f"gemm persistent {name} {M}x{N}x{K}", got, ref.gemm(x, w, bias, **kw))
        assert torch.equal(got, classic), f"persistent vs 128x128 kernel differ ({name})"
        assert torch.equal(got, pinned), f"persistent kernel with reserved CUs differs ({name})"
        # round 6: the direct (LDS-free) epilogue of the same kernel (A3D_GEMM_DIRECT): W rows staged in a permuted order, stores from the MFMA layout
        assert torch.equal(got, ops.gemm(x, w, bias, direct=True, **kw)), f"direct epilogue differs ({name})"
        assert torch.equal(ops.gemm(x, w, None, direct=True, **kw), ops.gemm(x, w, None, tile128=True, **kw)), f"direct epilogue differs without a bias ({name})"
    big = rnd(M, 3 * K, seed=16)
    xs = big[:, K:2 * K]
    out = torch.zeros(M, 2 * N, device="cuda", dtype=BF)
    ops.gemm(xs, w, bias, out=out[:, N:])
    check("gemm persistent strided", out[:, N:], ref.gemm(xs, w, bias))
    assert (out[:, :N] == 0).all()


# Round 6: small token matrices (M % 128 == 0, N % 128 == 0, K >= 256, too few 256-row tiles to fill the persistent kernel's grid) take the
# LDS-DMA ring kernel (csrc/gemm_ring.hip; 128 x 128 / 128 x 256 / 128 x 320 tiles chosen by plan_ring): same K order and epilogue arithmetic
# => bit-equal to the 128 x 128 kernel.  Shapes: 128 x 128 tiles, one per workgroup (80 - 160 tiles); 128 x 256 (M = 4096, N = 1280; 2048 x 3840;
# 1152 x 2560: a row count that is a multiple of 128 but not of 256); 128 x 320 (5120 x 960: 120 tiles; 38528 x 960: 903 tiles, three or four per
# workgroup — the K-tile stream and the bias parity run across tiles); more 128 x 128 tiles than CUs (40960 x 128); the shortest K, a long K
# (80 K-tiles through a ring of four), N = 128 (one column of tiles); 8192 x 1280 / 16384 x 640 are a toss-up between the persistent and the
# ring kernel (plan_ring's cost model currently keeps them on the persistent one).
@pytest.mark.parametrize("M,N,K", [(1024, 1280, 1280), (2048, 1280, 2560), (4096, 1280, 1280), (5120, 960, 320), (1024, 1280, 5120), (2048, 3840, 1280),
                                   (128, 128, 256), (384, 640, 256), (16384, 128, 640), (1152, 2560, 704), (8192, 1280, 640), (16384, 640, 320),
                                   (40960, 128, 256), (2048, 1280, 256), (38528, 960, 256)])
def test_gemm_ring_path(ops, ref, M, N, K):
    x, w = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=K ** -0.5)
    bias = rnd(N, seed=23, dtype=torch.float32)
    res = rnd(M, N, seed=24)
    rb = rnd(M // 128, N, seed=25)
    for name, kw in (("plain", {}), ("nobias", dict(nobias=True)), ("residual", dict(residual=res, alpha=0.37, beta=1.0)), ("residual beta", dict(residual=res, alpha=0.63, beta=0.9)),
                     ("rowbias+res", dict(rowbias=rb, rb_div=128, residual=res)), ("rowbias", dict(rowbias=rb, rb_div=128))):
        kw = dict(kw)
        b = None if kw.pop("nobias", False) else bias
        got, classic, pinned = _both_paths(ops, lambda t128: ops.gemm(x, w, b, tile128=t128, **kw))
        check(f"gemm ring {name} {M}x{N}x{K}", got, ref.gemm(x, w, b, **kw))
        assert torch.equal(got, classic), f"ring vs 128x128 kernel differ ({name})"
        assert torch.equal(got, pinned), f"ring kernel with reserved CUs differs ({name})"
    big = rnd(M, 3 * K, seed=26)
    xs = big[:, K:2 * K]                                       # strided A: a column block of a fused projection output (row stride 3 K)
    out = torch.zeros(M, 2 * N, device="cuda", dtype=BF)
    ops.gemm(xs, w, bias, out=out[:, N:])
    assert torch.equal(out[:, N:], ops.gemm(xs.contiguous(), w, bias, tile128=True))
    assert (out[:, :N] == 0).all()
    wrows = rnd(2 * N, K, seed=27, scale=K ** -0.5)            # a row slice of a fused weight (animate3d_amd.unet._rows)
    assert torch.equal(ops.gemm(x, wrows[N:], bias), ops.gemm(x, wrows[N:].contiguous(), bias, tile128=True))


def test_gemm_ring_path_repeatable_under_memory_traffic(ops):
    """As for the persistent kernel: counted vmcnt waits behind LDS-DMA loads; every launch must reproduce the 128 x 128 kernel bit for bit."""
    junk = torch.empty(32 << 20, device="cuda", dtype=torch.uint8)
    for (M, N, K) in [(1024, 1280, 1280), (4096, 1280, 2560), (2048, 640, 320), (8192, 640, 640)]:
        x, w = rnd(M, K, seed=45), rnd(N, K, seed=46, scale=K ** -0.5)
        bias, res = rnd(N, seed=47, dtype=torch.float32), rnd(M, N, seed=48)
        want = ops.gemm(x, w, bias, residual=res, alpha=0.7, tile128=True)
        for it in range(60):
            if it % 3 == 0:
                junk.add_(1)
            assert torch.equal(ops.gemm(x, w, bias, residual=res, alpha=0.7), want), (M, N, K, it)


def test_gemm_persistent_path_repeatable_under_memory_traffic(ops):
    """The persistent kernel keeps LDS-DMA loads and epilogue stores in flight behind hand-counted vmcnt waits; a miscount would
    show up as rare wrong tiles.  Hammer it next to unrelated HBM traffic: every launch must reproduce the 128x128 kernel bit for
    bit (2 200 launches over ten shapes were clean when this test was written)."""
    junk = torch.empty(32 << 20, device="cuda", dtype=torch.uint8)
    split, ops.split_k = ops.split_k, False           # (bit-for-bit against the 128x128 kernel: the unsplit persistent kernel)
    try:
        for (M, N, K) in [(65536, 320, 320), (32768, 1280, 1280), (8192, 1280, 1280)]:
            x, w = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=K ** -0.5)
            bias, res = rnd(N, seed=43, dtype=torch.float32), rnd(M, N, seed=44)
            want = ops.gemm(x, w, bias, residual=res, alpha=0.7, tile128=True)
            for it in range(40):
                if it % 3 == 0:
                    junk.add_(1)
                assert torch.equal(ops.gemm(x, w, bias, residual=res, alpha=0.7), want), (M, N, K, it)
    finally:
        ops.split_k = split


@pytest.mark.parametrize("M,N,K", [(8192, 1280, 1280), (2048, 3840, 1280), (2048, 1280, 5120), (8192, 1280, 2560), (4096, 640, 2304)])
def test_split_k_gemm(ops, ref, request, M, N, K):
    """Small M, long K (UNet levels 2 / 3; every GEMM of the 4D-SDS shape's mid block): a3d_gemm_ws deals the K-tiles of an output tile to
    several work items and a second kernel adds the fp32 slices in index order.  Against the fp32 reference at the usual bar, against the
    unsplit kernel to fp32 summation order (<= 1 storage ulp on a few elements), bit-reproducible run to run, every epilogue operand."""
    x, w = rnd(M, K, seed=51), rnd(N, K, seed=52, scale=K ** -0.5)
    bias, res, rb = rnd(N, seed=53, dtype=torch.float32), rnd(M, N, seed=54), rnd(M // 256, N, seed=55)
    ops.split_k_gemm = True            # (off by default: split-K pays for the mid-block convolutions, not for the level-2 / 3 linears)
    request.addfinalizer(lambda: setattr(ops, "split_k_gemm", False))
    for name, kw in (("plain", {}), ("residual", dict(residual=res, alpha=0.37, beta=0.9)), ("rowbias+res", dict(rowbias=rb, rb_div=256, residual=res))):
        got = ops.gemm(x, w, bias, **kw)
        check(f"split-K gemm {name} {M}x{N}x{K}", got, ref.gemm(x, w, bias, **kw))
        assert torch.equal(got, ops.gemm(x, w, bias, **kw)), "split-K must be deterministic"
        split, ops.split_k = ops.split_k, False
        try:
            unsplit = ops.gemm(x, w, bias, **kw)
        finally:
            ops.split_k = split
        d = (got.float() - unsplit.float()).abs()
        assert d.max().item() <= 2.0 ** -7 * max(1.0, unsplit.float().abs().max().item()) and (d > 0).float().mean().item() < 0.2, (name, d.max().item())
    assert any(v > 0 for k, v in ops._ws_plan.items() if k[0] == "gemm" and k[1:4] == (M, N, K)), "the shape was expected to split"


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(128, 8, 8, 1280, 1280), (32, 8, 8, 2560, 1280), (128, 4, 4, 1280, 1280), (16, 16, 16, 640, 640)])
def test_split_k_conv3x3(ops, ref, B, H, W, Cin, Cout):
    """The 8 x 8 / 4 x 4 mid-block convolutions (K = 9 Cin over <= 128 output tiles): work items hold whole 64-channel slices."""
    x, w = rnd(B * H * W, Cin, seed=61), rnd(Cout, 9 * Cin, seed=62, scale=(9 * Cin) ** -0.5)
    bias, res, rb = rnd(Cout, seed=63, dtype=torch.float32), rnd(B * H * W, Cout, seed=64), rnd(B * H * W // 256, Cout, seed=65)
    for name, kw in (("plain", {}), ("residual", dict(residual=res)), ("rowbias", dict(rowbias=rb, rb_div=256))):
        got, _, _ = ops.conv3x3(x, B, H, W, w, bias, **kw)
        want, _, _ = ref.conv3x3(x, B, H, W, w, bias, **kw)
        check(f"split-K conv {name} B{B} {H}x{W} {Cin}->{Cout}", got, want)
        assert torch.equal(got, ops.conv3x3(x, B, H, W, w, bias, **kw)[0])
        split, ops.split_k = ops.split_k, False
        try:
            unsplit, _, _ = ops.conv3x3(x, B, H, W, w, bias, **kw)
        finally:
            ops.split_k = split
        d = (got.float() - unsplit.float()).abs()
        assert d.max().item() <= 2.0 ** -7 * max(1.0, unsplit.float().abs().max().item()), (name, d.max().item())
    assert any(v > 0 for k, v in ops._ws_plan.items() if k[0] == "conv" and k[1:6] == (B, H, W, Cin, Cout)), "the shape was expected to split"


@pytest.mark.parametrize("M,N2,K", [(65536, 512, 64), (49152, 2560, 320)])
def test_gemm_geglu_persistent_path(ops, ref, M, N2, K):
    assert _persistent_eligible(M, N2, geglu=True)
    x, w = rnd(M, K, seed=21), rnd(N2, K, seed=22, scale=K ** -0.5)
    bias = rnd(N2, seed=23, dtype=torch.float32)
    w_il, b_il = ops.interleave_geglu(w), ops.interleave_geglu(bias)
    got, classic, pinned = _both_paths(ops, lambda t128: ops.gemm_geglu(x, w_il, b_il, tile128=t128))
    check(f"gemm_geglu persistent {M}x{N2}x{K}", got, ref.geglu(ref.gemm(x, w, bias)))
    assert torch.equal(got, classic) and torch.equal(got, pinned)


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [(16, 64, 64, 64, 320, 1, False), (64, 32, 32, 128, 320, 1, False), (64, 64, 64, 64, 320, 2, False),
                                                       (64, 32, 32, 64, 256, 1, False), (256, 16, 16, 192, 640, 1, False),
                                                       (16, 32, 32, 64, 320, 1, True), (64, 16, 16, 128, 256, 1, True), (4, 64, 128, 64, 320, 1, True)])
def test_conv3x3_persistent_path(ops, ref, B, H, W, Cin, Cout, stride, up):
    x, w = rnd(B * H * W, Cin, seed=31), rnd(Cout, 9 * Cin, seed=32, scale=(9 * Cin) ** -0.5)
    bias = rnd(Cout, seed=33, dtype=torch.float32)
    He, We = (2 * H, 2 * W) if up else (H, W)
    Ho, Wo = (He - 1) // stride + 1, (We - 1) // stride + 1
    assert _persistent_eligible(B * Ho * Wo, Cout)
    res = rnd(B * Ho * Wo, Cout, seed=34)
    rb = rnd(B, Cout, seed=35)
    for name, kw in (("plain", {}), ("rowbias", dict(rowbias=rb, rb_div=Ho * Wo)), ("residual", dict(residual=res))):
        (got, _, _), (classic, _, _), (pinned, _, _) = _both_paths(ops, lambda t128: ops.conv3x3(x, B, H, W, w, bias, stride=stride, up2x=up, tile128=t128, **kw))
        want, _, _ = ref.conv3x3(x, B, H, W, w, bias, stride=stride, up2x=up, **kw)
        check(f"conv persistent {name} B{B} {H}x{W} {Cin}->{Cout} s{stride} up{int(up)}", got, want)
        assert torch.equal(got, classic), f"persistent vs 128x128 conv differ ({name})"
        assert torch.equal(got, pinned)


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [(2, 8, 12, 64, 128, 1, False), (3, 8, 8, 128, 64, 2, False),
                                                      (2, 6, 4, 64, 320, 1, True), (1, 16, 16, 320, 4, 1, False),
                                                      (2, 5, 7, 64, 64, 2, False)])
def test_conv3x3(ops, ref, B, H, W, Cin, Cout, stride, up):
    x = rnd(B * H * W, Cin, seed=1)
    w = rnd(Cout, 9 * Cin, seed=2, scale=(9 * Cin) ** -0.5)
    bias = rnd(Cout, seed=3, dtype=torch.float32)
    y, Ho, Wo = ops.conv3x3(x, B, H, W, w, bias, stride=stride, up2x=up)
    yr, Hr, Wr = ref.conv3x3(x, B, H, W, w, bias, stride=stride, up2x=up)
    assert (Ho, Wo) == (Hr, Wr)
    check(f"conv {B}x{H}x{W} {Cin}->{Cout} s{stride} up{int(up)}", y, yr)


def test_conv3x3_epilogue(ops, ref):
    B, H, W, Cin, Cout = 4, 8, 8, 64, 128
    x, w = rnd(B * H * W, Cin, seed=1), rnd(Cout, 9 * Cin, seed=2, scale=(9 * Cin) ** -0.5)
    bias, rb, res = rnd(Cout, seed=3, dtype=torch.float32), rnd(2, Cout, seed=4), rnd(B * H * W, Cout, seed=5)
    y, _, _ = ops.conv3x3(x, B, H, W, w, bias, rowbias=rb, rb_div=2 * H * W, residual=res)
    yr, _, _ = ref.conv3x3(x, B, H, W, w, bias, rowbias=rb, rb_div=2 * H * W, residual=res)
    check("conv rowbias+residual", y, yr)


# ------------------------------------------------------------------ attention
# per-call kernel choices of a3d_flash_attn (flag bits of its last argument): default dispatch, the exact pass of the LDS-DMA staged
# kernels alone (A3D_ATTN_EXACT), the generic kernel of the short / ragged shapes (A3D_ATTN_PLAIN)
ATTN_MODES = {"default": {}, "exact": dict(exact=True), "plain": dict(plain=True)}


def _mv_maps(n, F, L):
    return RowMap(F, n * F * L, L, L, F * L), RowMap(F, n * F * L, 0, L, F * L)


@pytest.mark.parametrize("D", [40, 80, 160])
@pytest.mark.parametrize("b,n,F,L", [(1, 2, 3, 64), (2, 2, 2, 16), (1, 1, 2, 4), (1, 4, 2, 256), (1, 3, 2, 100)])
def test_flash_attn_multiview_and_first_frame(ops, ref, D, b, n, F, L):
    heads = 8
    C = heads * D
    rows = b * n * F * L
    qkv = rnd(rows, 3 * C, seed=D + L)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    qm, k0 = _mv_maps(n, F, L)
    S = n * L
    check(f"mv attn D{D} b{b} n{n} F{F} L{L}", ops.flash_attn(q, k, v, qm, qm, b * F, heads, S, S),
          ref.flash_attn(q, k, v, qm, qm, b * F, heads, S, S))
    check(f"i2v attn D{D} b{b} n{n} F{F} L{L}", ops.flash_attn(q, k, v, qm, k0, b * F, heads, S, S),
          ref.flash_attn(q, k, v, qm, k0, b * F, heads, S, S))


@pytest.mark.parametrize("D", [40, 80, 160])
def test_flash_attn_cross_text_ip(ops, ref, D):
    heads, V, F, L, T = 8, 2, 3, 64, 77
    C = heads * D
    q = rnd(V * F * L, C, seed=1)
    kvt, kvi = rnd(V * T, 2 * C, seed=2), rnd(V * 4, 2 * C, seed=3)
    qc = RowMap(1, L, 0, L, 0)
    o = ops.flash_attn(q, kvt[:, :C], kvt[:, C:], qc, RowMap(F, T, 0, T, 0), V * F, heads, L, T)
    o_r = ref.flash_attn(q, kvt[:, :C], kvt[:, C:], qc, RowMap(F, T, 0, T, 0), V * F, heads, L, T)
    check(f"text cross-attn D{D}", o, o_r)
    ops.flash_attn(q, kvi[:, :C], kvi[:, C:], qc, RowMap(F, 4, 0, 4, 0), V * F, heads, L, 4, out=o, out_scale=0.7, accumulate=True)
    o_r2 = ref.flash_attn(q, kvi[:, :C], kvi[:, C:], qc, RowMap(F, 4, 0, 4, 0), V * F, heads, L, 4, out=o_r.clone(), out_scale=0.7, accumulate=True)
    check(f"text+ip cross-attn D{D}", o, o_r2, tol=6e-3)


@pytest.mark.parametrize("D", [40, 80, 160])
@pytest.mark.parametrize("V,F,L", [(2, 3, 64), (1, 2, 300), (2, 2, 1024), (1, 1, 2100)])
def test_flash_attn_two_key_sets_in_one_launch(ops, ref, D, V, F, L):
    """a3d_flash_attn2: text tokens (77) + IP-Adapter image tokens (4) in one launch, each with its own softmax, against the two-call
    sequence of the fp32 reference (attention_processor.py:233, 254-283); head_dim 160 has no fused kernel and must say so (None).
    From 256 queries per group head_dim 40 takes the register-resident kernel of cross_attn.hip (ragged counts: 300 / 2 100 queries)."""
    heads, T, nt = 8, 77, 4
    C = heads * D
    q = rnd(V * F * L, C, seed=1)
    kvt, kvi = rnd(V * T, 2 * C, seed=2), rnd(V * nt, 2 * C, seed=3)
    qc = RowMap(1, L, 0, L, 0)
    got = ops.flash_attn2(q, kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], qc, RowMap(F, T, 0, T, 0), RowMap(F, nt, 0, nt, 0), V * F, heads,
                          L, T, nt, out_scale2=0.7)
    if D == 160:
        assert got is None
        return
    want = ref.flash_attn2(q, kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], qc, RowMap(F, T, 0, T, 0), RowMap(F, nt, 0, nt, 0), V * F, heads,
                           L, T, nt, out_scale2=0.7)
    check(f"text + ip attention in one launch D{D} V{V} F{F} L{L}", got, want)
    again = ops.flash_attn2(q, kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], qc, RowMap(F, T, 0, T, 0), RowMap(F, nt, 0, nt, 0), V * F, heads,
                            L, T, nt, out_scale2=0.7)
    assert torch.equal(got, again), "the launch must be bit-reproducible (an unpadded MFMA -> inline-asm hazard once made it drift by an ulp)"
    if D == 40 and L >= 256:      # other token counts through the same kernel: a full 96 + 32, single tokens
        for T2, nt2 in ((96, 32), (33, 1), (1, 5)):
            kvt2, kvi2 = rnd(V * T2, 2 * C, seed=4), rnd(V * nt2, 2 * C, seed=5)
            g2 = ops.flash_attn2(q, kvt2[:, :C], kvt2[:, C:], kvi2[:, :C], kvi2[:, C:], qc, RowMap(F, T2, 0, T2, 0), RowMap(F, nt2, 0, nt2, 0), V * F, heads,
                                 L, T2, nt2, out_scale2=1.3)
            w2 = ref.flash_attn2(q, kvt2[:, :C], kvt2[:, C:], kvi2[:, :C], kvi2[:, C:], qc, RowMap(F, T2, 0, T2, 0), RowMap(F, nt2, 0, nt2, 0), V * F, heads,
                                 L, T2, nt2, out_scale2=1.3)
            check(f"text + ip attention D{D} L{L} tokens {T2} + {nt2}", g2, w2)


def test_flash_attn_rescale_branch(ops, ref):
    """One key per tile far above the rest forces the running-max rescale at a chosen tile."""
    heads, D, L = 8, 40, 512
    C = heads * D
    q, k, v = rnd(L, C, seed=1), rnd(L, C, seed=2), rnd(L, C, seed=3)
    for t, row in enumerate((70, 200, 450)):
        k[row] = q[5 + t] * (4.0 + 2 * t)
    m = RowMap(1, L, 0, L, 0)
    check("attn rescale spikes", ops.flash_attn(q, k, v, m, m, 1, heads, L, L), ref.flash_attn(q, k, v, m, m, 1, heads, L, L))


@pytest.mark.parametrize("mode", ["default", "plain"])
@pytest.mark.parametrize("spikes", [(3,), (40, 70), (500,), (31, 32, 63, 64, 95, 96), (250, 260, 270, 280, 290, 300, 310)])
@pytest.mark.parametrize("L,q_len", [(512, 512), (1024, 700)])
def test_flash_attn_rescale_paths_at_level0_shapes(ops, ref, spikes, L, q_len, mode):
    """Level-0 shapes: the default dispatch (flash_attn_dm_kernel) and the plain exact kernel that serves every shape the LDS-DMA
    kernel does not (``plain=True``): its offset moves lazily per tile.  Spikes in the very first sub-tile (initial offset), in
    consecutive sub-tiles, at sub-tile borders and in the last one (peeled iterations); a ragged query count exercises the masked rows."""
    heads, D = 8, 40
    C = heads * D
    q, k, v = rnd(q_len, C, seed=1), rnd(L, C, seed=2), rnd(L, C, seed=3)
    for t, row in enumerate(spikes):
        k[row] = q[7 + 3 * t] * (3.0 + 1.5 * t)
    got = ops.flash_attn(q, k, v, RowMap(1, q_len, 0, q_len, 0), RowMap(1, L, 0, L, 0), 1, heads, q_len, L, **ATTN_MODES[mode])
    check(f"attn {mode} spikes {spikes} L{L} q{q_len}", got, ref.flash_attn(q, k, v, RowMap(1, q_len, 0, q_len, 0), RowMap(1, L, 0, L, 0), 1, heads, q_len, L))


@pytest.mark.parametrize("mode", ["default", "exact"])
@pytest.mark.parametrize("spikes,gain", [((), 1.0), ((3,), 3.0), ((40, 70), 3.0), ((500,), 4.0), ((31, 32, 63, 64, 95, 96), 3.0),
                                         ((250,), 12.0), ((100, 400), 40.0), ((-1,), 60.0)])
@pytest.mark.parametrize("L,q_len", [(512, 512), (1024, 700)])
def test_flash_attn_dma_kernel(ops, ref, mode, spikes, gain, L, q_len):
    """flash_attn_dm_kernel (LDS-DMA staging, dense LDS images, P·V through the 16x16x32 MFMA): the default = max-free pass (bf16: offset
    fixed after the first 32 keys) with the exact re-run on overflow, and its exact pass alone (``exact=True``).  Moderate spikes stay
    inside the max-free pass (probabilities up to ~2^70), gains >= 40 overflow bf16 and must take
    the re-run; a spike in the very last key, a ragged query count (masked rows) and spikes in the first sub-tile are covered.
    Bar: 4e-3 as for every attention kernel; 1e-2 for the gains >= 40, where scores reach ~370 log2 units and the 8-bit mantissa of
    the pre-scaled Q (a property of all the D = 40 kernels: Q' = bf16(Q * scale * log2 e)) moves near-ties between the huge scores."""
    heads, D = 8, 40
    C = heads * D
    q, k, v = rnd(q_len, C, seed=1), rnd(L, C, seed=2), rnd(L, C, seed=3)
    for t, row in enumerate(spikes):
        k[row % L] = q[7 + 3 * t] * (gain + 0.5 * t)
    qm, km = RowMap(1, q_len, 0, q_len, 0), RowMap(1, L, 0, L, 0)
    want = ref.flash_attn(q, k, v, qm, km, 1, heads, q_len, L)
    got = ops.flash_attn(q, k, v, qm, km, 1, heads, q_len, L, **ATTN_MODES[mode])
    check(f"dm attn {mode} spikes {spikes} x{gain} L{L} q{q_len}", got, want, tol=4e-3 if gain < 40 else 1e-2)


@pytest.mark.parametrize("mode", ["default", "exact"])
def test_flash_attn_dma_kernel_multiview_maps(ops, ref, mode):
    """The same kernel through the multi-view and first-frame row maps (segments of L rows, 64-key tiles wrap at segment ends),
    with accumulate / out_scale, against the fp32 reference."""
    heads, D, b, n, F, L = 8, 40, 2, 4, 2, 256
    C = heads * D
    qkv = rnd(b * n * F * L, 3 * C, seed=21)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    qm, k0 = _mv_maps(n, F, L)
    S = n * L
    kw = ATTN_MODES[mode]
    for km, nm in ((qm, "mv"), (k0, "i2v")):
        want = ref.flash_attn(q, k, v, qm, km, b * F, heads, S, S)
        check(f"dm {mode} {nm} attn", ops.flash_attn(q, k, v, qm, km, b * F, heads, S, S, **kw), want)
    base = rnd(b * n * F * L, C, seed=22)
    o = base.clone()
    ops.flash_attn(q, k, v, qm, k0, b * F, heads, S, S, out=o, out_scale=0.6, accumulate=True, **kw)
    o_r = ref.flash_attn(q, k, v, qm, k0, b * F, heads, S, S, out=base.float(), out_scale=0.6, accumulate=True)
    check(f"dm {mode} accumulate", o, o_r, tol=6e-3)


def test_flash_attn_d80_kernel_variants(ops, ref):
    """Head dim 80: the LDS-DMA staged kernel (default from 512 keys; its exact pass alone with ``exact=True``) and the generic
    kernels (``plain=True``: two query sub-tiles per wave from 2 048 tokens, one below) on long and ragged short shapes, each against
    the fp32 reference; the forced spikes exercise the rescale paths and (spike 2) the overflow re-run."""
    heads, D = 8, 80
    C = heads * D
    if True:
        for (n, F, L, spike) in [(4, 1, 512, 1), (3, 2, 100, 0), (4, 1, 512, 2), (4, 1, 128, 1)]:
            qkv = rnd(n * F * L, 3 * C, seed=L)
            q, k, v = qkv[:, :C].contiguous(), qkv[:, C:2 * C].contiguous(), qkv[:, 2 * C:].contiguous()
            if spike:
                k[700 % (n * F * L)] = q[9] * 4.0
                k[1500 % (n * F * L)] = q[11] * 6.0
            if spike == 2:          # ~260 log2 units above the rest: overflows the max-free pass (exact re-run)
                k[900] = q[13] * 20.0
            qm, k0 = _mv_maps(n, F, L)
            want = ref.flash_attn(q, k, v, qm, k0, F, heads, n * L, n * L)
            for mode, kw in ATTN_MODES.items():
                check(f"D80 n{n} L{L} spike{spike} {mode}", ops.flash_attn(q, k, v, qm, k0, F, heads, n * L, n * L, **kw), want,
                      tol=4e-3 if spike < 2 else 1e-2)


def test_flash_attn_kernel_variants_agree(ops, ref):
    """The default (LDS-DMA staged), its exact pass alone and the plain D = 40 kernel on one long multi-view shape, each against the fp32 reference."""
    heads, D, b, n, F, L = 8, 40, 1, 4, 2, 256
    C = heads * D
    qkv = rnd(b * n * F * L, 3 * C, seed=11)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    qm, k0 = _mv_maps(n, F, L)
    want = ref.flash_attn(q, k, v, qm, k0, b * F, heads, n * L, n * L)
    for mode, kw in ATTN_MODES.items():
        check(f"D40 kernel choice {mode}", ops.flash_attn(q, k, v, qm, k0, b * F, heads, n * L, n * L, **kw), want)


@pytest.mark.parametrize("D", [40, 80, 160])
@pytest.mark.parametrize("V,F,L", [(2, 3, 16), (1, 4, 64), (2, 16, 64), (1, 32, 8), (1, 3, 5), (3, 16, 7)])
def test_temporal_attn(ops, ref, D, V, F, L):
    heads = 8
    C = heads * D
    qkv = rnd(V * F * L, 3 * C, seed=F + D)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    check(f"temporal attn D{D} V{V} F{F} L{L}", ops.temporal_attn(q, k, v, V, F, L, heads), ref.temporal_attn(q, k, v, V, F, L, heads))


@pytest.mark.parametrize("D,V,F,shards,L", [(40, 2, 16, 4, 64), (80, 1, 4, 2, 7), (160, 2, 32, 8, 8), (40, 3, 6, 3, 5)])
def test_temporal_attn_frame_sharded(ops, ref, D, V, F, shards, L):
    """a3d_temporal_attn_sharded_bf16: every frame shard's output from the rank-major all-gathered K|V must equal its rows of the
    unsharded kernel's output bit for bit (same per-query arithmetic) and the fp32 reference within the attention bar."""
    heads = 8
    C = heads * D
    qkv = rnd(V * F * L, 3 * C, seed=F + D)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    whole = ops.temporal_attn(q, k, v, V, F, L, heads)
    check(f"temporal attn D{D} (whole)", whole, ref.temporal_attn(q, k, v, V, F, L, heads))
    fl = F // shards
    blocks = lambda t: t.reshape(V, shards, fl, L, -1).permute(1, 0, 2, 3, 4).reshape(V * F * L, -1).contiguous()   # what all_gather_into_tensor builds
    kv_all = blocks(qkv[:, C:])                                  # [shards, (v f_l) l, 2C]
    for r in range(shards):
        q_loc = q.reshape(V, shards, fl, L, C)[:, r].reshape(V * fl * L, C).contiguous()
        got = ops.temporal_attn(q_loc, kv_all[:, :C], kv_all[:, C:], V, F, L, heads, q_f0=r * fl, q_frames=fl)
        want = whole.reshape(V, shards, fl, L, C)[:, r].reshape(V * fl * L, C)
        assert torch.equal(got, want), (r, (got.float() - want.float()).abs().max().item())
        check(f"temporal attn D{D} shard {r}/{shards} vs fp32", got, ref.temporal_attn(q_loc, kv_all[:, :C], kv_all[:, C:], V, F, L, heads, q_f0=r * fl, q_frames=fl))


# ------------------------------------------------------------------ normalisation
@pytest.mark.parametrize("B,rows,C,shards", [(2, 4 * 64, 320, 4), (3, 2 * 100, 640, 2), (1, 8 * 16, 1280, 8)])
def test_group_norm_split_halves(ops, ref, B, rows, C, shards):
    """a3d_group_norm_sums_bf16 + a3d_group_norm_apply_bf16: partial sums of row shards, summed, must reproduce the fused
    kernel (the frame-sharded 3-D GroupNorm of the motion modules)."""
    x = rnd(B * rows, C, seed=C) + 0.5
    gamma = 1 + 0.1 * rnd(C, seed=1, dtype=torch.float32)
    beta = 0.1 * rnd(C, seed=2, dtype=torch.float32)
    fused = ops.group_norm(x, B, rows, gamma, beta, 32, 1e-6, False)
    rl = rows // shards
    parts = [x.reshape(B, shards, rl, C)[:, r].reshape(B * rl, C).contiguous() for r in range(shards)]
    sums = sum(ops.group_norm_sums(p_, B, rl, 32) for p_ in parts)
    want_sums = ref.group_norm_sums(x, B, rows, 32)
    torch.testing.assert_close(sums, want_sums, rtol=1e-6, atol=1e-6)
    cnt = float(rows * (C // 32))
    mean = sums[..., 0] / cnt
    var = (sums[..., 1] / cnt - mean * mean).clamp_min(0.0)
    stats = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-6)], dim=-1).float().contiguous()
    for r, p_ in enumerate(parts):
        got = ops.group_norm_apply(p_, B, rl, gamma, beta, 32, stats, False)
        want = fused.reshape(B, shards, rl, C)[:, r].reshape(B * rl, C)
        check(f"group_norm split shard {r}/{shards} C{C}", got, want, tol=4e-3, max_ulps=2)

# round 6: instances that fit a workgroup's registers (norms.hip: gn_fused_kernel, <= 21 16-byte chunks per thread) take the one-launch kernel —
# every shape of the first line, 2000 x 1280 and 1024 x 640 (21 chunks per thread, 512 threads), 1024 x 320 (BASELINE config 5's level 0);
# 4096 x 320 / 4096 x 1280 / 2100 x 640 stay on the three-launch path (statistics workspace)
@pytest.mark.parametrize("B,rows,C", [(3, 64, 320), (2, 300, 640), (2, 256, 960), (1, 1000, 1280), (2, 64, 1920), (2, 16, 2560), (1, 4, 1280),
                                      (1, 2000, 1280), (2, 1024, 640), (3, 1024, 320), (2, 4096, 320), (1, 4096, 1280), (2, 2100, 640), (9, 256, 1280)])
@pytest.mark.parametrize("silu", [False, True])
def test_group_norm(ops, ref, B, rows, C, silu):
    x = rnd(B * rows, C, seed=C) + 0.5
    gamma = 1 + 0.1 * rnd(C, seed=1, dtype=torch.float32)
    beta = 0.1 * rnd(C, seed=2, dtype=torch.float32)
    check(f"group_norm B{B} rows{rows} C{C} silu{int(silu)}", ops.group_norm(x, B, rows, gamma, beta, 32, 1e-5, silu),
          ref.group_norm(x, B, rows, gamma, beta, 32, 1e-5, silu), tol=4e-3, max_ulps=3)


@pytest.mark.parametrize("rows,C", [(256, 1280), (1024, 640), (64, 1920), (4096, 320)])
def test_group_norm_does_not_depend_on_the_batch(ops, rows, C):
    """An instance's result must not depend on how many instances the launch holds (a CFG- / view-sharded rank normalises a slice of the
    batch and has to reproduce the unsharded job bit for bit): one-launch and three-launch paths alike."""
    x = rnd(5 * rows, C, seed=C + rows) + 0.25
    gamma = 1 + 0.1 * rnd(C, seed=1, dtype=torch.float32)
    beta = 0.1 * rnd(C, seed=2, dtype=torch.float32)
    full = ops.group_norm(x, 5, rows, gamma, beta, 32, 1e-5, True)
    for b in (0, 3):
        one = ops.group_norm(x[b * rows:(b + 1) * rows].contiguous(), 1, rows, gamma, beta, 32, 1e-5, True)
        assert torch.equal(one, full[b * rows:(b + 1) * rows])
    two = ops.group_norm(x[rows:3 * rows].contiguous(), 2, rows, gamma, beta, 32, 1e-5, True)
    assert torch.equal(two, full[rows:3 * rows])


# M >= 4096 with C in {320, 640, 1280} takes the sub-wave-rows kernel (ragged last batch: M not a multiple of the rows per block)
@pytest.mark.parametrize("M,C", [(10, 320), (257, 640), (64, 1280), (8, 768), (4096, 320), (5003, 320), (4101, 640), (4099, 1280), (6000, 768)])
def test_layer_norm(ops, ref, M, C):
    x = rnd(M, C, seed=C) * 2 + 0.3
    gamma, beta = 1 + 0.1 * rnd(C, seed=1, dtype=torch.float32), 0.1 * rnd(C, seed=2, dtype=torch.float32)
    check(f"layer_norm {M}x{C}", ops.layer_norm(x, gamma, beta, 1e-5), ref.layer_norm(x, gamma, beta, 1e-5), max_ulps=3)
    pe1, pe2 = rnd(3, C, seed=3), rnd(5, C, seed=4)
    y1, y2 = ops.layer_norm(x, gamma, beta, 1e-5, pe1=pe1, pe1_div=4, pe2=pe2, pe2_div=1, two=True)
    r1, r2 = ref.layer_norm(x, gamma, beta, 1e-5, pe1=pe1, pe1_div=4, pe2=pe2, pe2_div=1, two=True)
    check(f"layer_norm+pe1 {M}x{C}", y1, r1, max_ulps=3)
    check(f"layer_norm+pe2 {M}x{C}", y2, r2, max_ulps=3)


# ------------------------------------------------------------------ elementwise / layout
def test_geglu_silu_concat(ops, ref):
    x = rnd(333, 2560, seed=1) * 2
    check("geglu", ops.geglu(x), ref.geglu(x), max_ulps=3)
    s = rnd(40, 1280, seed=2) * 3
    check("silu", ops.silu(s), ref.silu(s), max_ulps=3)
    a, b = rnd(100, 640, seed=3), rnd(100, 320, seed=4)
    assert torch.equal(ops.concat(a, b), torch.cat([a, b], 1))


def test_timestep_embed(ops, ref):
    t = torch.tensor([0.0, 1.0, 501.0, 961.0, 999.0], device="cuda")
    check("timestep", ops.timestep_embed(t, 320), ref.timestep_embed(t, 320), tol=3e-3, max_ulps=3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_im2col_and_unpack(ops, ref, dtype):
    s = rnd(2, 4, 3, 8, 16, seed=1, dtype=dtype)
    got, want = ops.im2col_in(s), ref.im2col_in(s).to(BF)
    assert torch.equal(got, want)
    x = rnd(2 * 3 * 8 * 16, 4, seed=2)
    assert torch.equal(ops.unpack_out(x, 2, 4, 3, 8, 16, dtype), ref.unpack_out(x, 2, 4, 3, 8, 16, dtype))


def test_cfg_ddim_step(ops, ref):
    n, C, F, H, W = 2, 4, 3, 8, 8
    eps, x = rnd(2 * n, C, F, H, W, seed=1, dtype=torch.float32), rnd(n, C, F, H, W, seed=2, dtype=torch.float32)
    first = rnd(n, C, 1, H, W, seed=3, dtype=torch.float32)
    got = ops.cfg_ddim_step(eps, x, first, 7.5, 0.31, 0.42)
    want = ref.cfg_ddim_step(eps, x.clone(), first, 7.5, 0.31, 0.42)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_bad_arguments_raise(ops):
    with pytest.raises(RuntimeError):
        ops.gemm(rnd(8, 100), rnd(16, 100))                    # K % 64 != 0
    with pytest.raises(RuntimeError):
        ops.gemm(rnd(8, 64, dtype=torch.float32), rnd(16, 64))  # not bf16
    q = rnd(64, 8 * 48)
    with pytest.raises(RuntimeError):
        ops.flash_attn(q, q, q, RowMap(1, 64, 0, 64, 0), RowMap(1, 64, 0, 64, 0), 1, 8, 64, 64)   # head_dim 48


# ------------------------------------------------------------------ fp16 storage (the a3d_*_f16 twins)
@pytest.fixture(scope="module")
def ops16():
    from animate3d_amd.hip_ops import HipOps
    return HipOps(act_dtype=torch.float16)


H16 = torch.float16


def test_fp16_storage_gemm_conv(ops16, ref):
    """Same kernels compiled with IEEE fp16 as the storage type (v_mfma_f32_32x32x16_f16, fp32 accumulate).  The reference is
    fp32 arithmetic on the SAME fp16 inputs, so what differs is the summation order and the final fp16 rounding (2^-11 = 4.9e-4):
    bar 1e-3 relative L2 (the bf16 bar is 4e-3)."""
    for (M, N, K) in [(300, 320, 320), (4096, 1280, 320), (389, 1280, 640), (130, 2560, 1280)]:
        x, w = rnd(M, K, seed=1, dtype=H16), rnd(N, K, seed=2, scale=K ** -0.5, dtype=H16)
        bias, res = rnd(N, seed=3, dtype=torch.float32), rnd(M, N, seed=4, dtype=H16)
        check(f"f16 gemm {M}x{N}x{K}", ops16.gemm(x, w, bias, residual=res, alpha=0.7), ref.gemm(x, w, bias, residual=res, alpha=0.7), tol=1e-3)
    x, w = rnd(2048, 320, seed=5, dtype=H16), rnd(2560, 320, seed=6, scale=320 ** -0.5, dtype=H16)
    bias = rnd(2560, seed=7, dtype=torch.float32)
    il = ops16.interleave_geglu
    check("f16 gemm+geglu", ops16.gemm_geglu(x, il(w), il(bias)), ref.gemm_geglu(x, il(w), il(bias)), tol=1e-3)
    for (B, H, W, Cin, Cout, st, up) in [(2, 8, 12, 64, 128, 1, False), (3, 8, 8, 128, 64, 2, False), (2, 6, 4, 64, 320, 1, True), (4, 16, 16, 320, 320, 1, False)]:
        xc, wc = rnd(B * H * W, Cin, seed=1, dtype=H16), rnd(Cout, 9 * Cin, seed=2, scale=(9 * Cin) ** -0.5, dtype=H16)
        bc = rnd(Cout, seed=3, dtype=torch.float32)
        y, Ho, Wo = ops16.conv3x3(xc, B, H, W, wc, bc, stride=st, up2x=up)
        yr, _, _ = ref.conv3x3(xc, B, H, W, wc, bc, stride=st, up2x=up)
        check(f"f16 conv {B}x{H}x{W} {Cin}->{Cout} s{st} up{int(up)}", y, yr, tol=1e-3)


@pytest.mark.parametrize("D", [40, 80, 160])
def test_fp16_storage_attention(ops16, ref, D):
    heads = 8
    C = heads * D
    for (b, n, F, L) in [(1, 2, 3, 64), (1, 4, 2, 256), (1, 3, 2, 100)]:          # 4 x 256: the interleaved kernel at D = 40
        qkv = rnd(b * n * F * L, 3 * C, seed=D + L, dtype=H16)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        qm, k0 = _mv_maps(n, F, L)
        S = n * L
        check(f"f16 mv attn D{D} n{n} F{F} L{L}", ops16.flash_attn(q, k, v, qm, qm, b * F, heads, S, S), ref.flash_attn(q, k, v, qm, qm, b * F, heads, S, S), tol=1.5e-3)
        check(f"f16 i2v attn D{D} n{n} F{F} L{L}", ops16.flash_attn(q, k, v, qm, k0, b * F, heads, S, S), ref.flash_attn(q, k, v, qm, k0, b * F, heads, S, S), tol=1.5e-3)
    if D == 40:                                  # rescale path of the interleaved kernel with fp16 offsets / probabilities
        L = 512
        q, k, v = rnd(L, C, seed=1, dtype=H16), rnd(L, C, seed=2, dtype=H16), rnd(L, C, seed=3, dtype=H16)
        for t, row in enumerate((3, 70, 200, 450)):
            k[row] = q[5 + t] * (3.0 + 1.5 * t)
        m = RowMap(1, L, 0, L, 0)
        check("f16 attn rescale spikes", ops16.flash_attn(q, k, v, m, m, 1, heads, L, L), ref.flash_attn(q, k, v, m, m, 1, heads, L, L), tol=1.5e-3)
    V, F, L = 2, 16, 64
    qkv = rnd(V * F * L, 3 * C, seed=F + D, dtype=H16)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    check(f"f16 temporal attn D{D}", ops16.temporal_attn(q, k, v, V, F, L, heads), ref.temporal_attn(q, k, v, V, F, L, heads), tol=1.5e-3)


@pytest.mark.parametrize("mode", ["default", "exact"])
@pytest.mark.parametrize("spikes,gain", [((), 1.0), ((3,), 1.5), ((40, 70), 1.5), ((250,), 2.2), ((500,), 4.0), ((-1,), 12.0), ((16, 48, 80), 2.0)])
@pytest.mark.parametrize("L,q_len", [(512, 512), (1024, 700)])
def test_fp16_flash_attn_dma_kernel(ops16, ref, mode, spikes, gain, L, q_len):
    """fp16 storage through flash_attn_dm_kernel: the max-free pass (default) estimates the offset from 32 keys spread over the
    key range (rows 0, L/32, 2L/32, ...) and keeps P inside fp16's 2^-24 .. 2^16: spikes of 9-14 log2 units above the sample stay inside
    the window (gain 1.5 - 2.2: also when the spike IS a sample key), larger ones (gain >= 4) overflow to inf and must take the exact
    re-run; ``exact=True`` is the exact pass alone.  Bar: the fp16 attention bar (1.5e-3)."""
    heads, D = 8, 40
    C = heads * D
    q, k, v = rnd(q_len, C, seed=1, dtype=H16), rnd(L, C, seed=2, dtype=H16), rnd(L, C, seed=3, dtype=H16)
    for t, row in enumerate(spikes):
        k[row % L] = q[7 + 3 * t] * (gain + 0.25 * t)
    qm, km = RowMap(1, q_len, 0, q_len, 0), RowMap(1, L, 0, L, 0)
    want = ref.flash_attn(q, k, v, qm, km, 1, heads, q_len, L)
    got = ops16.flash_attn(q, k, v, qm, km, 1, heads, q_len, L, **ATTN_MODES[mode])
    check(f"f16 dm attn {mode} spikes {spikes} x{gain} L{L} q{q_len}", got, want, tol=1.5e-3)


def test_fp16_flash_attn_dma_kernels_multiview_maps(ops16, ref):
    """fp16 storage, default dispatch (LDS-DMA kernels at head_dim 40 and 80, sampled max-free pass) through the multi-view and
    first-frame row maps — the sample keys are spread over all the views' segments —, against the fp32 reference and against the
    generic exact kernels (``plain=True``)."""
    heads, b, n, F = 8, 2, 4, 2
    if True:
        for D, L in ((40, 256), (80, 256)):
            C = heads * D
            qkv = rnd(b * n * F * L, 3 * C, seed=21 + D, dtype=H16)
            q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
            qm, k0 = _mv_maps(n, F, L)
            S = n * L
            for km, nm in ((qm, "mv"), (k0, "i2v")):
                want = ref.flash_attn(q, k, v, qm, km, b * F, heads, S, S)
                got = ops16.flash_attn(q, k, v, qm, km, b * F, heads, S, S)
                check(f"f16 dm D{D} {nm} attn", got, want, tol=1.5e-3)
                check(f"f16 dm D{D} {nm} attn vs plain kernel", got, ops16.flash_attn(q, k, v, qm, km, b * F, heads, S, S, plain=True), tol=1.5e-3)
            q2, k2, v2 = q.contiguous(), k.clone(), v.contiguous()
            k2[5 * L + 17] = q2[9] * 9.0         # far outside the fp16 window: exact re-run of the workgroups that see it
            check(f"f16 dm D{D} overflow re-run", ops16.flash_attn(q2, k2, v2, qm, qm, b * F, heads, S, S), ref.flash_attn(q2, k2, v2, qm, qm, b * F, heads, S, S), tol=1.5e-3)
            qw, kw = (q2.float() * 2.5).to(H16), (k2.float() * 2.5).to(H16)      # scores 6 x wider: the sample's spread sends every workgroup to the exact pass
            check(f"f16 dm D{D} wide scores", ops16.flash_attn(qw, kw, v2, qm, qm, b * F, heads, S, S), ref.flash_attn(qw, kw, v2, qm, qm, b * F, heads, S, S), tol=1.5e-3)


def test_fp16_storage_norms_and_elementwise(ops16, ref):
    x = rnd(2 * 300, 640, seed=3, dtype=H16) + 0.5
    gamma, beta = 1 + 0.1 * rnd(640, seed=1, dtype=torch.float32), 0.1 * rnd(640, seed=2, dtype=torch.float32)
    ulp = lambda got, want, n: (got.float() - want.float()).abs().max().item() <= n * want.float().abs().max().item() * 2.0 ** -10
    for silu in (False, True):
        got, want = ops16.group_norm(x, 2, 300, gamma, beta, 32, 1e-5, silu), ref.group_norm(x, 2, 300, gamma, beta, 32, 1e-5, silu)
        assert ulp(got, want, 3), ("group_norm", silu)
    got, want = ops16.layer_norm(x, gamma, beta, 1e-5), ref.layer_norm(x, gamma, beta, 1e-5)
    assert ulp(got, want, 3)
    pe1, pe2 = rnd(3, 640, seed=3, dtype=H16), rnd(5, 640, seed=4, dtype=H16)
    y1, y2 = ops16.layer_norm(x, gamma, beta, 1e-5, pe1=pe1, pe1_div=4, pe2=pe2, pe2_div=1, two=True)
    r1, r2 = ref.layer_norm(x, gamma, beta, 1e-5, pe1=pe1, pe1_div=4, pe2=pe2, pe2_div=1, two=True)
    assert ulp(y1, r1, 3) and ulp(y2, r2, 3)
    u = rnd(100, 2560, seed=5, dtype=H16)
    assert ulp(ops16.geglu(u), ref.geglu(u), 3) and ulp(ops16.silu(u), ref.silu(u), 3)
    a, b = rnd(64, 640, seed=6, dtype=H16), rnd(64, 320, seed=7, dtype=H16)
    assert torch.equal(ops16.concat(a, b), torch.cat([a, b], 1))
    t = torch.tensor([1.0, 501.0, 999.0], device="cuda")
    assert ulp(ops16.timestep_embed(t, 320), ref.timestep_embed(t, 320), 3)
    for src in (torch.float32, torch.bfloat16, torch.float16):
        smp = rnd(2, 4, 3, 8, 8, seed=8, dtype=src)
        rows = ops16.im2col_in(smp)
        assert rows.dtype == H16 and ulp(rows, ref.im2col_in(smp), 2)
        xr = rnd(2 * 3 * 8 * 8, 4, seed=9, dtype=H16)
        out = ops16.unpack_out(xr, 2, 4, 3, 8, 8, src)
        assert out.dtype == src and ulp(out, ref.unpack_out(xr, 2, 4, 3, 8, 8, src), 3)


@pytest.mark.parametrize("M,Ka,Kb,N", [(32768, 1280, 640, 640), (65536, 640, 320, 320), (8192, 1280, 1280, 1280), (65536, 320, 320, 320)])
def test_gemm_two_source_a_operand(ops, ref, M, Ka, Kb, N):
    """a3d_gemm2: [xa | xb] w^T + bias read from the two parts (the 1x1 shortcut conv of an up-block ResNet over cat([hidden, skip], 1),
    unet_motion_mv_model.py:826-827) — same K order as a3d_gemm on the concatenated operand, so BIT-identical to it; a shape the persistent
    kernel does not take returns None (the caller concatenates)."""
    xa, xb = rnd(M, Ka, seed=71), rnd(M, Kb, seed=72)
    w, bias = rnd(N, Ka + Kb, seed=73, scale=(Ka + Kb) ** -0.5), rnd(N, seed=74, dtype=torch.float32)
    cat = ops.concat(xa, xb)
    want = ops.gemm(cat, w, bias)
    got = ops.gemm2(xa, xb, w, bias)
    assert got is not None, "expected the persistent kernel to take this shape"
    check(f"gemm2 {M}x{N}x({Ka}+{Kb})", got, ref.gemm(cat, w, bias))
    assert torch.equal(got, want), "two-source GEMM differs from the GEMM over the concatenated operand"
    assert torch.equal(ops.gemm2(xa, xb, w, None), ops.gemm(cat, w, None))
    assert ops.gemm2(xa[:300], xb[:300], w, bias) is None          # M % 256 != 0: not the persistent kernel's -> caller falls back
    if N == 320:
        assert ops.gemm2(xa[:16384], xb[:16384], w, bias) is None  # 64 tiles on 256 CUs: under the persistent kernel's fill threshold


@pytest.mark.parametrize("B,rows,Ca,Cb", [(8, 1024, 1280, 640), (16, 4096, 640, 320), (4, 256, 1280, 1280), (8, 4096, 320, 320)])
@pytest.mark.parametrize("silu", [True, False])
def test_group_norm_two_source(ops, ref, B, rows, Ca, Cb, silu):
    """a3d_group_norm2: GroupNorm(+SiLU) of cat([xa, xb], 1) read from its parts — groups that straddle the boundary included (1280 + 640
    channels: 32 groups of 60) — bit-identical to a3d_group_norm of the concatenated tensor (same partial sums in the same order)."""
    xa, xb = rnd(B * rows, Ca, seed=81), rnd(B * rows, Cb, seed=82)
    C = Ca + Cb
    gamma, beta = rnd(C, seed=83, dtype=torch.float32), rnd(C, seed=84, dtype=torch.float32)
    cat = ops.concat(xa, xb)
    want = ops.group_norm(cat, B, rows, gamma, beta, 32, 1e-5, silu)
    got = ops.group_norm2(xa, xb, B, rows, gamma, beta, 32, 1e-5, silu)
    check(f"group_norm2 B{B} rows{rows} {Ca}+{Cb} silu={silu}", got, ref.group_norm(cat, B, rows, gamma, beta, 32, 1e-5, silu))
    assert torch.equal(got, want)
