"""CPU restatement of the fp16 max-free softmax of csrc/flash_attn_dm.hip / flash_attn_dm80.hip (DM_BIAS, DM_VAR_MAX): what the window
costs in accuracy and when the kernel leaves it.

The kernel fixes one offset per query before the key loop — maximum of 32 sample scores (keys 0, kv/32, 2 kv/32, ...) + 4 log2 units —,
stores P = 2^(s - offset) in fp16 (2^-24 .. 65504, subnormals kept) and accumulates P and P·V in fp32.  A query predicts an overflow when
the variance of its sample scores exceeds 28; a workgroup (512 queries at head_dim 40) skips the max-free pass when MORE THAN A QUARTER of
its queries do (round 5: a vote — the sample variance scatters by +-25 %, an OR sent every workgroup to the exact pass at score spreads
well inside the window) and re-runs exactly when a row sum comes out non-finite.  This
file checks the arithmetic of that plan on synthetic score rows, in numpy-like torch on the CPU; the GPU tests
(tests/test_hip_kernels_gpu.py::test_fp16_flash_attn_dma_kernel*) check the kernels."""
import math

import pytest
import torch

BIAS, VAR_MAX = 4.0, 28.0


def _window_attention(s, v):
    """s [q, k] scores in log2 units (fp32), v [k, d]: returns (out fp32 [q, d], row_sum, wide [q], overflow [q])."""
    k = s.shape[1]
    sample = s[:, :: k // 32][:, :32]
    off = (sample.max(dim=1).values + BIAS).to(torch.float16).float()           # round16: the offset rides in a 16-bit operand slot
    wide = sample.var(dim=1, unbiased=False) > VAR_MAX
    p = torch.exp2(s - off[:, None]).to(torch.float16)                          # inf above 65504, subnormals down to 2^-24, 0 below
    l = p.float().sum(dim=1)
    overflow = ~torch.isfinite(l)
    out = (p.float() @ v.to(torch.float16).float()) / l[:, None]
    return out, l, wide, overflow


def _exact(s, v):
    w = torch.softmax(s.double() * math.log(2.0), dim=1)
    return (w @ v.to(torch.float16).double()).float()


@pytest.mark.parametrize("sd", [0.5, 1.5, 3.0, 4.5])
@pytest.mark.parametrize("k", [1024, 16384])
def test_rows_inside_the_window_lose_nothing(sd, k):
    """Gaussian scores up to the predictor's threshold (sd^2 <= 28 would be sd <= 5.3; 4.5 keeps the sample estimate of the variance under
    it for most rows): no row overflows, and the result is the fp16 answer — error vs the exact softmax is the 2^-11 rounding of P."""
    g = torch.Generator().manual_seed(int(sd * 10) + k)
    s = torch.randn(256, k, generator=g) * sd
    v = torch.randn(k, 40, generator=g)
    out, l, wide, overflow = _window_attention(s, v)
    assert not overflow.any()
    keep = ~wide
    assert keep.float().mean() > (0.95 if sd < 4.5 else 0.5)
    want = _exact(s, v)
    err = float((out[keep] - want[keep]).norm() / want[keep].norm())
    print(f"[parity] fp16 window sd={sd} k={k}: rel L2 {err:.2e}, rows sent to the exact pass by the predictor {float(wide.float().mean()):.3f}")
    assert err <= 6e-4
    assert float(l[keep].min()) >= 2.0 ** -BIAS * 0.99         # the sample maximum itself contributes 2^-4: the row sum can never vanish


def test_peaked_rows_are_caught_by_the_predictor_or_by_the_row_sum():
    """Scores 4 x wider than the window tolerates (sd = 12): the sample variance flags almost every row before the pass starts; a row that
    slips through and overflows shows up as a non-finite row sum.  No row may be BOTH accepted and wrong."""
    g = torch.Generator().manual_seed(5)
    s = torch.randn(512, 4096, generator=g) * 12.0
    v = torch.randn(4096, 40, generator=g)
    out, l, wide, overflow = _window_attention(s, v)
    assert float(wide.float().mean()) > 0.97
    accepted = ~wide & ~overflow
    if accepted.any():
        want = _exact(s, v)
        assert float((out[accepted] - want[accepted]).norm() / want[accepted].norm()) <= 2e-3


def test_single_spike_outside_the_sample_overflows_loudly():
    """One key 30 log2 units above everything else, not among the sample keys, calm scores otherwise (the predictor sees nothing): P = inf,
    the row sum is not finite -> the kernel's exact re-run.  The same spike 15 units up stays inside the window and is exact to rounding."""
    g = torch.Generator().manual_seed(6)
    s = torch.randn(64, 4096, generator=g)
    v = torch.randn(4096, 40, generator=g)
    hot = s.clone(); hot[:, 1001] += 30.0
    _, l, wide, overflow = _window_attention(hot, v)
    assert not wide.any() and overflow.all()
    warm = s.clone(); warm[:, 1001] += 15.0
    out, l, wide, overflow = _window_attention(warm, v)
    assert not wide.any() and not overflow.any()
    assert float((out - _exact(warm, v)).norm() / _exact(warm, v).norm()) <= 6e-4


def test_flat_background_under_a_spike_keeps_its_mass():
    """The case that decides the bias: a spike INSIDE the sample (so the offset sits 4 above it) over a flat background 2^-11 of its
    height with 16 384 keys carries 8 x the spike's mass; fp16 keeps 2^-11 * 2^-4 = 2^-15 as a normal-range number (subnormals start at
    2^-14: one bit lost), so the background survives — with a bias of 14 (P_max = 2^-14) it would be flushed to 2^-24 steps and lost."""
    k = 16384
    s = torch.full((4, k), -11.0)
    s[:, 0] = 0.0                                # key 0 is a sample key
    v = torch.randn(k, 40, generator=torch.Generator().manual_seed(7))
    out, l, wide, overflow = _window_attention(s, v)
    want = _exact(s, v)
    assert not overflow.any()
    assert float((out - want).norm() / want.norm()) <= 2e-3


@pytest.mark.parametrize("sd_nat,expect_exact", [(1.0, False), (3.0, False), (3.4, False), (4.5, True), (6.0, True)])
def test_workgroup_vote_follows_the_true_spread(sd_nat, expect_exact):
    """The workgroup-level decision of the kernels (``__syncthreads_count(wide) * 4 > threads``) on 512 Gaussian rows of 16 384 keys with
    natural-log score sd ``sd_nat`` (what tools/microbench.py flashspread sweeps): up to sd 3.4 (variance 24 in log2 units, threshold 28)
    the max-free pass runs although a few per cent of the rows read above the threshold; from sd 4.5 the workgroup goes exact right away.
    Rows of an accepted workgroup that do overflow are caught by their row sum (non-finite), never silently wrong."""
    g = torch.Generator().manual_seed(int(sd_nat * 100))
    s = torch.randn(512, 16384, generator=g) * (sd_nat / math.log(2.0))
    v = torch.randn(16384, 40, generator=g)
    out, l, wide, overflow = _window_attention(s, v)
    frac = float(wide.float().mean())
    goes_exact = frac * 4 > 1.0
    print(f"[parity] fp16 window, score sd {sd_nat} (natural log): {100 * frac:.1f} % of the rows predict an overflow, "
          f"{100 * float(overflow.float().mean()):.2f} % overflow in the max-free pass -> workgroup takes the {'exact' if goes_exact else 'max-free'} pass")
    assert goes_exact == expect_exact
    if not goes_exact:
        ok = ~overflow
        want = _exact(s, v)
        assert float((out[ok] - want[ok]).norm() / want[ok].norm()) <= 6e-4          # accepted rows are right whatever their own predictor said
        assert float(overflow.float().mean()) <= 0.01                                # and a re-run for an overflowing row stays rare
