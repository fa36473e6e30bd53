#!/usr/bin/env python
"""In-process A/B micro-benchmarks of the dominant kernels on one MI355X (HIP events on the launch stream,
interleaved rounds, median reported).  Usage:  python tools/microbench.py [flash] [gemm] [conv] [misc]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import animate3d_amd.hip_ops as _hip_ops  # noqa: E402
from animate3d_amd.hip_ops import HipOps, RowMap  # noqa: E402

if os.environ.get("A3D_LIB"):        # A/B of two builds of the library (this tool only; the package always loads the in-tree one)
    _hip_ops._LIB_PATH = os.environ["A3D_LIB"]

BF = torch.bfloat16


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device="cuda", dtype=torch.float32) * scale).to(BF)


def timeit(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts), min(ts)


def graph_time(fn, n=20, reps=7):
    """GPU time per call without host gaps: n consecutive calls captured in a HIP graph, replayed; median over reps (us)."""
    fn(); fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return statistics.median(ts)


ATTN_MODES = (("default", {}), ("exact", dict(exact=True)), ("plain", dict(plain=True)))


def bench_flash(ops, shapes=((40, 4, 16, 4096, 2), (80, 4, 16, 1024, 2), (160, 4, 16, 256, 2))):
    print("== flash attention (multi-view map): default dispatch | exact pass of the LDS-DMA kernels | generic kernel; median ms / TFLOP/s; err = rel L2 vs default")
    for (D, n, F, L, b) in shapes:
        heads, C = 8, 8 * D
        rows = b * n * F * L
        qkv = rnd(rows, 3 * C)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        qm = RowMap(F, n * F * L, L, L, F * L)
        S, G = n * L, b * F
        flops = 4.0 * G * S * S * C
        ref = ops.flash_attn(q, k, v, qm, qm, G, heads, S, S).float()
        for name, kw in ATTN_MODES * 2:
            out = ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, **kw).float()
            err = ((out - ref).norm() / ref.norm()).item()
            med, mn = timeit(lambda: ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, **kw), reps=5 if D == 40 else 10)
            print(f"D={D:3d} S={S:5d} G={G} {name:8s}: {med:8.3f} ms  {flops / med / 1e9:7.1f} TF/s (best {flops / mn / 1e9:7.1f})  err={err:.2e}")


def bench_flashdt(_ops):
    """bf16 against fp16 storage on the LDS-DMA kernels at short and long key counts (in-graph timing): what fp16's sampled-offset prologue
    (one extra 32-key sample tile, its MFMAs, the statistics and the workgroup vote) costs per workgroup."""
    for (D, n, F, L, b) in ((40, 4, 16, 1024, 1), (40, 4, 16, 4096, 1), (40, 4, 16, 256, 1), (80, 4, 16, 256, 1), (80, 4, 16, 1024, 1)):
        heads, C = 8, 8 * D
        rows = b * n * F * L
        qm = RowMap(F, n * F * L, L, L, F * L)
        S, G = n * L, b * F
        flops = 4.0 * G * S * S * C
        line = f"D={D:3d} S={S:5d} G={G}:"
        for dt in (torch.bfloat16, torch.float16):
            ops = HipOps(act_dtype=dt)
            q, k, v = (torch.randn(rows, C, device="cuda").to(dt) for _ in range(3))
            for name, kw in (("default", {}), ("exact", dict(exact=True))):
                us = graph_time(lambda: ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, **kw), n=20 if S <= 4096 else 3)
                line += f"  {str(dt)[6:]} {name} {us:8.1f} us {flops / us / 1e6:6.1f} TF/s |"
        print(line, flush=True)


def bench_flashrank(ops):
    """The level-0 attention launch as a rank of a multi-GPU layout sees it (its own views' queries against the gathered keys of all views:
    q_len = views_local x L, kv_len = 4 L) next to the single-GPU launch, same keys per group: time per workgroup round should not differ."""
    D, n, F, L = 40, 4, 16, 4096
    heads, C = 8, 8 * D
    k, v = rnd(n * F * L, C), rnd(n * F * L, C)
    km = RowMap(F, n * F * L, L, L, F * L)
    for n_loc in (4, 2, 1):
        q = rnd(n_loc * F * L, C)
        qm = RowMap(F, n_loc * F * L, L, L, F * L)
        S_q, S_kv, G = n_loc * L, n * L, F
        flops = 4.0 * G * S_q * S_kv * C
        for rep in range(2):
            us = graph_time(lambda: ops.flash_attn(q, k, v, qm, km, G, heads, S_q, S_kv), n=4)
            wgs = heads * (S_q // 512) * G
            print(f"D=40 G={G} q_len={S_q:5d} kv_len={S_kv}: {us:9.1f} us  {flops / us / 1e6:7.1f} TF/s  | {wgs} workgroups = {wgs / 256:.0f} rounds of 256: {us / (wgs / 256):7.1f} us per round", flush=True)


def bench_flashdm(ops, scales=(1.0, 0.0, 3.0)):
    """Level-0 launch shape of BASELINE config 2 (32 groups x 8 heads x 16384 x 16384, head_dim 40): the LDS-DMA staged kernel (default), its exact
    pass alone and the generic kernel, interleaved rounds in one process.  Input scales: randn, zeros (clock ceiling), randn x 3 (peaky scores)."""
    D, n, F, L, b = 40, 4, 16, 4096, 2
    heads, C = 8, 8 * D
    qm = RowMap(F, n * F * L, L, L, F * L)
    S, G = n * L, b * F
    flops = 4.0 * G * S * S * C
    for sc in scales:
        qkv = rnd(b * n * F * L, 3 * C, scale=sc)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        ref = ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, plain=True).float()
        for name, kw in ATTN_MODES * 2:
            out = ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, **kw).float()
            err = ((out - ref).norm() / (ref.norm() + 1e-30)).item()
            med, mn = timeit(lambda: ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, **kw), reps=5)
            print(f"level-0 D=40 scale={sc} {name:8s}: {med:8.3f} ms  {flops / med / 1e9:7.1f} TF/s (best {flops / mn / 1e9:7.1f})  err vs generic = {err:.2e}", flush=True)


def bench_flashspread(_ops, sds=(0.3, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0)):
    """The LDS-DMA attention kernels (head_dim 40 at the level-0 launch shape, head_dim 80 at the level-1 shape) on scores of a given
    spread: q, k ~ N(0, sd) so that scale * q.k has standard deviation sd (natural-log units, what softmax sees); v ~ N(0, 1).  Per
    storage type: the default dispatch (max-free pass, exact re-run per workgroup when its row sums leave the window — fp16 storage also when the
    sampled variance predicts it), the exact pass alone, and the share of the exact pass in the default launch estimated from the two
    (t_default(sd) - t_default(0.3)) / t_exact(sd).  bench.py's synthetic weights give sd ~ 0.3; trained attention layers sit at 1-4."""
    for dt in (torch.bfloat16, torch.float16):
        ops = HipOps(act_dtype=dt)
        for (D, n, F, L, b) in ((40, 4, 16, 4096, 2), (80, 4, 16, 1024, 2)):
            heads, C = 8, 8 * D
            rows = b * n * F * L
            qm = RowMap(F, n * F * L, L, L, F * L)
            S, G = n * L, b * F
            flops = 4.0 * G * S * S * C
            base = None
            for sd in sds:
                q = (torch.randn(rows, C, device="cuda") * sd ** 0.5).to(dt)
                k = (torch.randn(rows, C, device="cuda") * sd ** 0.5).to(dt)
                v = torch.randn(rows, C, device="cuda").to(dt)
                ref = ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, plain=True).float()
                cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
                ops.attn_counters = cnt                   # a3d_flash_attn_counted: [0] voted exact, [1] re-ran after an overflow, [2] workgroups
                out = ops.flash_attn(q, k, v, qm, qm, G, heads, S, S).float()
                ops.attn_counters = None
                voted, rerun, launched = (int(x) for x in cnt[:3].tolist())
                err = ((out - ref).norm() / (ref.norm() + 1e-30)).item()
                reps = 5 if D == 40 else 11
                t_def, _ = timeit(lambda: ops.flash_attn(q, k, v, qm, qm, G, heads, S, S), reps=reps)
                t_ex, _ = timeit(lambda: ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, exact=True), reps=reps)
                base = t_def if base is None else base
                share = min(1.0, max(0.0, (t_def - base) / t_ex))
                print(f"{str(dt)[6:]:8s} D={D:3d} S={S:5d} score sd={sd:3.1f}: default {t_def:7.3f} ms {flops / t_def / 1e9:7.1f} TF/s | exact pass only {t_ex:7.3f} ms | "
                      f"exact-pass share of the default launch ~{share:4.2f} | workgroups {launched}: voted exact {voted}, re-ran after overflow {rerun} | "
                      f"err vs generic kernel {err:.2e}", flush=True)


def bench_gemm(ops):
    print("== GEMM  Y[M,N] = X[M,K] W[N,K]^T (+bias +residual); median ms / TFLOP/s;  default dispatch | 128 x 128-tile kernel")
    shapes = [(524288, 1280, 320), (524288, 320, 320), (524288, 2560, 320), (524288, 320, 1280), (524288, 960, 320),
              (131072, 2560, 640), (131072, 640, 640), (131072, 5120, 640), (131072, 640, 2560),
              (32768, 5120, 1280), (32768, 1280, 1280), (32768, 10240, 1280), (32768, 1280, 5120), (8192, 1280, 1280), (8192, 10240, 1280)]
    for (M, N, K) in shapes:
        x, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device="cuda")
        res = rnd(M, N)
        fl = 2.0 * M * N * K
        out = []
        for t128 in (False, True):
            med, mn = timeit(lambda: ops.gemm(x, w, bias, residual=res, tile128=t128), reps=7)
            med2, _ = timeit(lambda: ops.gemm(x, w, bias, tile128=t128), reps=7)
            out.append(f"{med:7.3f} ms {fl / med / 1e9:6.1f} TF/s | no-residual {med2:7.3f} ms {fl / med2 / 1e9:6.1f} TF/s")
        print(f"M={M:7d} N={N:5d} K={K:5d}: " + "  ||  ".join(out), flush=True)
    print("-- fused GEGLU projection")
    for (M, N2, K) in [(524288, 2560, 320), (131072, 5120, 640), (32768, 10240, 1280)]:
        x, w = rnd(M, K), rnd(N2, K, scale=K ** -0.5)
        bias = torch.randn(N2, device="cuda")
        fl = 2.0 * M * N2 * K
        out = []
        for t128 in (False, True):
            med, mn = timeit(lambda: ops.gemm_geglu(x, w, bias, tile128=t128), reps=7)
            out.append(f"{med:7.3f} ms {fl / med / 1e9:6.1f} TF/s")
        print(f"M={M:7d} N2={N2:5d} K={K:5d}: " + "  ||  ".join(out), flush=True)


def bench_conv(ops):
    print("== conv3x3 NHWC implicit GEMM; median ms / TFLOP/s")
    shapes = [(128, 64, 64, 320, 320, 1, False), (128, 64, 64, 960, 320, 1, False), (128, 64, 64, 640, 320, 1, False),
              (128, 32, 32, 640, 640, 1, False), (128, 32, 32, 1920, 640, 1, False), (128, 16, 16, 1280, 1280, 1, False),
              (128, 16, 16, 2560, 1280, 1, False), (128, 8, 8, 2560, 1280, 1, False), (128, 64, 64, 320, 320, 2, False),
              (128, 32, 32, 640, 640, 1, True), (128, 64, 64, 320, 4, 1, False)]
    for (B, H, W, Cin, Cout, st, up) in shapes:
        x, w = rnd(B * H * W, Cin), rnd(Cout, 9 * Cin, scale=(9 * Cin) ** -0.5)
        bias = torch.randn(Cout, device="cuda")
        He, We = (2 * H, 2 * W) if up else (H, W)
        Ho, Wo = (He - 1) // st + 1, (We - 1) // st + 1
        fl = 2.0 * B * Ho * Wo * 9 * Cin * Cout
        out = []
        for t128 in (False, True):
            med, mn = timeit(lambda: ops.conv3x3(x, B, H, W, w, bias, stride=st, up2x=up, tile128=t128), reps=5)
            out.append(f"{'tile128' if t128 else 'default'}: {med:7.3f} ms {fl / med / 1e9:6.1f} TF/s")
        print(f"B={B} {H}x{W} {Cin:4d}->{Cout:4d} s{st} up{int(up)}: " + "  ||  ".join(out))


def bench_smallm(ops):
    """The small-M shapes of a multi-GPU rank / BASELINE config 5: level 2 / 3 linears (M = 1024 ... 16384), GroupNorm
    instances of a few hundred rows, mid-block convolutions of 16 images.  Linears: default dispatch | 128 x 128 kernel | split-K."""
    print("== small-M linears: default | tile128 | split-K (HipOps.split_k_gemm); median ms / TFLOP/s")
    shapes = [(1024, 1280, 1280, 0), (1024, 1280, 2560, 1), (1024, 1280, 5120, 1), (1024, 3840, 1280, 0), (1024, 10240, 1280, 0),
              (2048, 1280, 1280, 0), (2048, 1280, 2560, 1), (2048, 1280, 5120, 1), (2048, 3840, 1280, 0),
              (4096, 1280, 1280, 0), (4096, 1280, 2560, 1), (4096, 1280, 5120, 1), (4096, 2560, 1280, 0), (4096, 3840, 1280, 0),
              (8192, 1280, 1280, 1), (8192, 1280, 2560, 1), (8192, 1280, 5120, 1), (8192, 3840, 1280, 0),
              (16384, 640, 640, 1), (16384, 640, 1280, 1), (16384, 640, 2560, 1), (16384, 1920, 640, 0), (16384, 2560, 1280, 0),
              (32768, 640, 640, 1), (65536, 320, 320, 1), (65536, 960, 320, 0), (131072, 320, 320, 1)]
    for (M, N, K, r) in shapes:
        x, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device="cuda")
        res = rnd(M, N) if r else None
        fl = 2.0 * M * N * K
        out = []
        for name, kw, sk in (("default", {}, False), ("tile128", dict(tile128=True), False), ("split-K", {}, True)):
            ops.split_k_gemm = sk
            med, mn = timeit(lambda: ops.gemm(x, w, bias, residual=res, **kw), reps=15, warm=3)
            gt = graph_time(lambda: ops.gemm(x, w, bias, residual=res, **kw))
            out.append(f"{name} {med * 1e3:6.1f} us eager, {gt:6.1f} us in a graph = {fl / gt / 1e6:6.1f} TF/s")
        ops.split_k_gemm = False
        print(f"M={M:6d} N={N:5d} K={K:5d}{' +res' if r else '     '}: " + "  |  ".join(out), flush=True)
    bench_gn(ops)
    print("== mid-block convolutions of a rank of 8 (B = 16 images): split-K on | off")
    for (B, H, W, Cin, Cout) in [(16, 8, 8, 1280, 1280), (16, 8, 8, 2560, 1280), (16, 16, 16, 1280, 1280), (16, 16, 16, 2560, 1280), (16, 32, 32, 640, 640),
                                 (32, 8, 8, 1280, 1280), (32, 16, 16, 1280, 1280), (128, 4, 4, 1280, 1280)]:
        x, w = rnd(B * H * W, Cin), rnd(Cout, 9 * Cin, scale=(9 * Cin) ** -0.5)
        bias = torch.randn(Cout, device="cuda")
        fl = 2.0 * B * H * W * 9 * Cin * Cout
        out = []
        for sk in (True, False):
            ops.split_k = sk
            med, mn = timeit(lambda: ops.conv3x3(x, B, H, W, w, bias), reps=15, warm=3)
            out.append(f"split_k={sk}: {med * 1e3:7.1f} us {fl / med / 1e9:6.1f} TF/s")
        ops.split_k = True
        print(f"conv3x3 B={B:3d} {H}x{W} {Cin}->{Cout}: " + "  |  ".join(out), flush=True)


def bench_shortk(ops):
    """In-graph time of the short-K linears / GEGLU projections of levels 0 / 1 on the default dispatch (A/B of side builds: A3D_LIB=...)."""
    import os
    print(f"== short-K shapes, default dispatch, in-graph us / TFLOP/s  (library: {os.environ.get('A3D_LIB', 'shipped')})")
    for (M, N, K, r, geglu) in [(524288, 960, 320, 0, 0), (524288, 1280, 320, 0, 0), (524288, 2560, 320, 0, 1), (524288, 320, 320, 1, 0), (524288, 320, 640, 1, 0),
                                (524288, 320, 1280, 1, 0), (131072, 1920, 640, 0, 0), (131072, 5120, 640, 0, 1), (131072, 640, 640, 1, 0), (131072, 640, 2560, 1, 0),
                                (32768, 3840, 1280, 0, 0), (32768, 10240, 1280, 0, 1), (32768, 1280, 5120, 1, 0)]:
        x, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device="cuda")
        res = rnd(M, N) if r else None
        fl = 2.0 * M * N * K
        if geglu:
            wi, bi = ops.interleave_geglu(w), ops.interleave_geglu(bias)
            fn = lambda: ops.gemm_geglu(x, wi, bi)
        else:
            fn = lambda: ops.gemm(x, w, bias, residual=res)
        ts = [graph_time(fn, n=10) for _ in range(3)]
        print(f"{'geglu' if geglu else 'gemm '} M={M:6d} N={N:5d} K={K:5d}{' +res' if r else '     '}: " + "  ".join(f"{t:7.1f}" for t in ts) + f" us   {fl / min(ts) / 1e6:7.1f} TF/s", flush=True)


def bench_onewave(ops):
    """VERDICT r5 item 4b, first half: the one-wave-per-SIMD GEMM (gemm_ring.hip with 128 x 320 tiles: 256 threads, one wave per SIMD, serial
    epilogue) against the shipped two-waves-per-SIMD ping-pong kernel (256 x 320 tiles) on the epilogue-bound short-K shapes of level 0 / 1,
    in-graph GPU time.  The ring kernel has no second accumulator set: if its main loop + serial epilogue is far behind here, hiding the
    epilogue cannot win the 8 % the experiment was asked to show."""
    print("== short-K linears at the level-0 / 1 shapes: persistent ping-pong kernel (default) | one wave per SIMD, 128 x 320 tiles (ring=True); in-graph us / TFLOP/s")
    for (M, N, K, r) in [(524288, 960, 320, 0), (524288, 1280, 320, 0), (524288, 320, 320, 1), (524288, 320, 1280, 1), (131072, 1920, 640, 0),
                         (131072, 640, 640, 1), (131072, 640, 2560, 1), (32768, 1280, 1280, 0), (32768, 1280, 5120, 1)]:
        x, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device="cuda")
        res = rnd(M, N) if r else None
        fl = 2.0 * M * N * K
        out = []
        for name, kw in (("ping-pong 256x320", {}), ("one-wave 128x320", dict(ring=True))) * 2:
            gt = graph_time(lambda: ops.gemm(x, w, bias, residual=res, **kw), n=10)
            out.append(f"{name} {gt:7.1f} us {fl / gt / 1e6:7.1f} TF/s")
        assert torch.equal(ops.gemm(x, w, bias, residual=res), ops.gemm(x, w, bias, residual=res, ring=True))
        print(f"M={M:6d} N={N:5d} K={K:5d}{' +res' if r else '     '}: " + "  |  ".join(out), flush=True)


def bench_directepi(ops):
    """the persistent kernel with the direct epilogue (W rows permuted on the DMA source address so that a lane holds 16
    consecutive columns of one output row: two 16-byte stores per lane and 32 x 32 tile, bias / rowbias / residual read in that layout, no LDS
    transposition) against the shipped LDS-transposing epilogue.  In-graph GPU time, alternating; results must be bit-identical."""
    print("== persistent GEMM: LDS-transposing epilogue (shipped) | direct epilogue (direct=True); in-graph us / TFLOP/s")
    for (M, N, K, r, b) in [(524288, 1280, 320, 0, 1), (524288, 960, 320, 0, 0), (524288, 960, 320, 0, 1), (524288, 320, 320, 1, 1), (524288, 320, 640, 1, 1), (524288, 320, 1280, 1, 1),
                            (131072, 1920, 640, 0, 0), (131072, 640, 640, 1, 1), (131072, 640, 2560, 1, 1), (131072, 2560, 640, 0, 1),
                            (32768, 1280, 1280, 0, 1), (32768, 3840, 1280, 0, 0), (32768, 1280, 5120, 1, 1)]:
        x, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device="cuda") if b else None
        res = rnd(M, N) if r else None
        fl = 2.0 * M * N * K
        out = []
        for name, kw in (("transposing", {}), ("direct", dict(direct=True))) * 2:
            gt = graph_time(lambda: ops.gemm(x, w, bias, residual=res, **kw), n=10)
            out.append(f"{name} {gt:7.1f} us {fl / gt / 1e6:7.1f} TF/s")
        same = torch.equal(ops.gemm(x, w, bias, residual=res), ops.gemm(x, w, bias, residual=res, direct=True))
        print(f"M={M:6d} N={N:5d} K={K:5d}{' +res' if r else '     '}{' +bias' if b else '      '}: " + "  |  ".join(out) + f"  | bit-identical: {same}", flush=True)


def bench_gn(ops):
    print("== GroupNorm (+SiLU) instances: B x rows x C; median us / GB/s (algorithmic: read + write)")
    for (B, rows, C) in [(16, 256, 1280), (16, 64, 1280), (1, 1024, 1280), (1, 4096, 1280), (16, 1024, 640), (16, 4096, 320), (32, 256, 1280), (32, 64, 1280),
                         (128, 256, 1280), (128, 64, 1280), (128, 1024, 640), (128, 4096, 320), (8, 4096, 1280), (8, 1024, 1280), (16, 256, 2560), (16, 256, 1920),
                         (64, 256, 1280), (64, 1024, 640), (64, 64, 1280), (128, 1024, 320), (128, 256, 640), (128, 64, 2560), (128, 256, 1920), (128, 1024, 960)]:
        x = rnd(B * rows, C)
        gam, bet = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        med, mn = timeit(lambda: ops.group_norm(x, B, rows, gam, bet, 32, 1e-5, True), reps=15, warm=3)
        gt = graph_time(lambda: ops.group_norm(x, B, rows, gam, bet, 32, 1e-5, True))
        print(f"group_norm B={B:3d} rows={rows:5d} C={C:4d}: {med * 1e3:7.1f} us eager, {gt:7.1f} us in a graph = {2.0 * x.numel() * 2 / gt / 1e3:7.0f} GB/s", flush=True)


def bench_gemmscale(ops):
    """Per-K-tile time of the persistent GEMM against the number of active CUs: G output tiles of 256 x 320 (one per workgroup),
    K = 5120 (80 K-tiles per tile), so the time of a launch is 80 x the K-tile period (+ one epilogue).  Flat in G = latency /
    issue bound per CU; growing with G = shared bandwidth (L2 / fabric) bound."""
    print("== persistent GEMM: period per 64 of K vs active CUs (M = 256 G, N = 320, K = 5120)")
    for G in (1, 32, 64, 128, 256, 1024):
        x = rnd(256 * G, 5120)
        w = rnd(320, 5120, scale=0.02)
        med, mn = timeit(lambda: ops.gemm(x, w), reps=7, warm=2)
        rounds = (G + 255) // 256
        print(f"G={G:4d}: {med * 1e3:8.1f} us   K-tile period {mn * 1e3 / (80 * rounds):6.3f} us  ({2.0 * 256 * G * 320 * 5120 / mn / 1e9:7.1f} TF/s best)")


def bench_gemmcal(ops):
    """Calibration against the CDNA4 guide's verified plain-HIP 256^2 8-phase template (1 320-1 340 TFLOP/s at 4096^3, ~1 470 at 8192^3 on
    uniform random [-1, 1) bf16 operands; 1 563 / 1 728 on zero-filled ones): the shipped GEMM on the same shapes and the same data fills."""
    print("== GEMM calibration: square shapes, uniform random [-1,1) | zero-filled operands; median ms / TFLOP/s (best)")
    for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (8192, 5120, 8192), (16384, 3840, 4096), (32768, 1280, 5120), (32768, 3840, 1280)]:
        outs = []
        for fill in ("uniform", "zeros"):
            if fill == "uniform":
                x = (torch.rand(M, K, device="cuda") * 2 - 1).to(BF); w = (torch.rand(N, K, device="cuda") * 2 - 1).to(BF)
            else:
                x = torch.zeros(M, K, device="cuda", dtype=BF); w = torch.zeros(N, K, device="cuda", dtype=BF)
            fl = 2.0 * M * N * K
            med, mn = timeit(lambda: ops.gemm(x, w), reps=15, warm=3)
            outs.append(f"{fill}: {med:7.3f} ms {fl / med / 1e9:7.1f} TF/s (best {fl / mn / 1e9:7.1f})")
        print(f"M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(outs), flush=True)


def bench_flash16(_ops):
    """fp16 storage: LDS-DMA kernels with the sampled max-free pass (default) against their exact pass and the generic kernels, BASELINE config-2 launch shapes."""
    ops = HipOps(act_dtype=torch.float16)
    print("== fp16 storage flash attention: default | exact | generic; median ms / TFLOP/s; err = rel L2 vs generic kernel")
    for (D, n, F, L, b) in ((40, 4, 16, 4096, 2), (80, 4, 16, 1024, 2)):
        heads, C = 8, 8 * D
        rows = b * n * F * L
        for scale in (1.0, 2.0):
            qkv = (torch.randn(rows, 3 * C, device="cuda") * scale).to(torch.float16)
            q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
            qm = RowMap(F, n * F * L, L, L, F * L)
            S, G = n * L, b * F
            flops = 4.0 * G * S * S * C
            ref = ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, plain=True).float()
            for name, kw in ATTN_MODES * 2:
                out = ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, **kw).float()
                err = ((out - ref).norm() / ref.norm()).item()
                med, mn = timeit(lambda: ops.flash_attn(q, k, v, qm, qm, G, heads, S, S, **kw), reps=5 if D == 40 else 10)
                print(f"f16 D={D:3d} S={S:5d} scale={scale} {name:8s}: {med:8.3f} ms  {flops / med / 1e9:7.1f} TF/s (best {flops / mn / 1e9:7.1f})  err={err:.2e}")


def bench_wgrad(ops):
    """Weight gradient dW = dY^T X at the train.yaml shapes (4 views x 16 frames x 32x32 latent)."""
    print("== wgrad (LDS-DMA + ds_read_b64_tr_b16): median us / TFLOP/s")
    for (M, N, K) in [(65536, 320, 320), (65536, 2560, 320), (65536, 320, 1280), (16384, 640, 640), (16384, 5120, 640), (16384, 640, 2560),
                      (4096, 1280, 1280), (4096, 10240, 1280), (4096, 1280, 5120), (1024, 1280, 1280)]:
        dy, x = rnd(M, N), rnd(M, K)
        fl = 2.0 * M * N * K
        med, mn = timeit(lambda: ops.wgrad(dy, x), reps=9, warm=2)
        print(f"M={M:6d} N={N:5d} K={K:5d}: {med * 1e3:7.1f} us {fl / med / 1e9:6.1f} TF/s")


def bench_attnbwd(ops):
    """Attention backward at the train.yaml level-0 / level-1 shapes (4 views x 16 frames x 32x32 latent): dQ + dK|dV with the forward's
    statistics (a3d_attn_delta + the two gradient passes) and with the statistics pass; multi-view and first-frame maps."""
    print("== attention backward: median ms (with forward statistics | recomputing them)")
    for (D, n, F, L, b) in ((40, 4, 16, 1024, 1), (80, 4, 16, 256, 1), (160, 4, 16, 64, 1)):
        heads, C = 8, 8 * D
        rows = b * n * F * L
        kvq = rnd(rows, 3 * C)
        q, k, v = kvq[:, 2 * C:], kvq[:, :C], kvq[:, C:2 * C]
        do = rnd(rows, C)
        qm = RowMap(F, n * F * L, L, L, F * L)
        k0 = RowMap(F, n * F * L, 0, L, F * L)
        S, G = n * L, b * F
        for name, km, share in (("multi-view", qm, 1), ("first-frame", k0, F)):
            o, lse = ops.flash_attn(q, k, v, qm, km, G, heads, S, S, with_lse=True)
            fast, _ = timeit(lambda: ops.flash_attn_bwd(q, k, v, do, qm, km, G, heads, S, S, q_per_kv=share, o=o, lse=lse), reps=9, warm=2)
            slow, _ = timeit(lambda: ops.flash_attn_bwd(q, k, v, do, qm, km, G, heads, S, S, q_per_kv=share), reps=9, warm=2)
            fwd, _ = timeit(lambda: ops.flash_attn(q, k, v, qm, km, G, heads, S, S), reps=9, warm=2)
            fl = 4.0 * G * S * S * C
            print(f"D={D:3d} S={S:5d} {name:11s}: {fast:7.3f} ms ({2.5 * fl / fast / 1e9:6.1f} TF/s) | {slow:7.3f} ms   forward {fwd:6.3f} ms ({fl / fwd / 1e9:6.1f} TF/s)")


def bench_graph(ops):
    """Eager launches vs HIP-graph replay of one denoise step at the small BASELINE configurations."""
    import time
    from animate3d_amd.config import UNetConfig
    from animate3d_amd.unet import MVUNetMotionModel
    sys.path.insert(0, ROOT)
    from bench import make_inputs
    cfg = UNetConfig()
    model = MVUNetMotionModel(cfg, num_views=4, device="cuda")
    model.init_synthetic(seed=0)
    model = model.to(torch.bfloat16).eval()
    print("== one denoise step, eager launches vs HIP-graph replay (ms)")
    for tag, V, n, F, lat in (("config 5 (SDS: 4 views x 16 frames, 32x32 latent, CFG)", 8, 4, 16, 32), ("config 1 (1 view x 4 frames, 64x64 latent)", 1, 1, 4, 64)):
        model.num_views = n
        inp = make_inputs(cfg, V, n, F, (lat, lat), torch.device("cuda"))
        def run(fn, reps=5):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        eager = run(lambda: model(**inp))
        step = model.capture_graph(**inp)
        graph = run(lambda: step(**inp))
        print(f"{tag}: eager {eager:8.2f}   graph {graph:8.2f}")


def bench_loop(ops):
    """BASELINE config 2 end to end: 25 DDIM steps (UNet forward on the CFG-doubled batch + fused CFG/DDIM/re-pin) on one GPU."""
    import time
    from animate3d_amd.config import UNetConfig
    from animate3d_amd.denoise import denoise_loop
    from animate3d_amd.embeddings import get_camera
    from animate3d_amd.unet import MVUNetMotionModel
    cfg = UNetConfig()
    n, F, hw = 4, 16, (64, 64)
    model = MVUNetMotionModel(cfg, num_views=n, device="cuda")
    model.init_synthetic(seed=0)
    model = model.to(torch.bfloat16).eval()
    g = torch.Generator().manual_seed(1)
    first = (0.18215 * torch.randn(n, 4, 1, *hw, generator=g)).cuda()
    latents = torch.cat([first, torch.randn(n, 4, F - 1, *hw, generator=g).cuda()], dim=2)
    pe = torch.randn(2 * n, 77, cfg.cross_attention_dim, generator=g).cuda()
    ie = torch.randn(2 * n, cfg.ip_image_embed_dim, generator=g).cuda(); ie[:n] = 0
    cam = get_camera(n).cuda()
    denoise_loop(model, latents, first, pe, ie, cam, num_inference_steps=2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = denoise_loop(model, latents, first, pe, ie, cam, num_inference_steps=25)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"== config 2, 25 DDIM steps (4 views x 16 frames x 64x64 latent, CFG 7.5): {dt:.2f} s = {dt / 25 * 1e3:.1f} ms per step; finite={bool(torch.isfinite(out).all())}")


def bench_vae(ops):
    """decode_latents of BASELINE config 2's output: 4 views x 16 frames of 64x64 latents -> 64 images of 512x512."""
    import time
    from animate3d_amd.vae import AutoencoderKLDecoder
    vae = AutoencoderKLDecoder(device="cuda").init_synthetic(seed=0).to(torch.bfloat16).eval()
    lat = torch.randn(4, 4, 16, 64, 64, device="cuda") * 0.18215
    vae.decode_latents(lat[:, :, :2]); torch.cuda.synchronize()
    for frames in (4, 16):
        x = lat[:, :, :frames].contiguous()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = vae.decode_latents(x)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        n = 4 * frames
        from animate3d_amd.flops import vae_decode_flops
        fl = vae_decode_flops(64, 64) / 1e12
        print(f"== VAE decode {n} images 64x64 latent -> 512x512: {dt * 1e3:8.1f} ms = {dt / n * 1e3:6.2f} ms per image "
              f"({fl * n / dt:6.1f} TFLOP/s at {fl:.2f} TFLOP per image); out {tuple(y.shape)} finite={bool(torch.isfinite(y).all())} "
              f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


def bench_misc(ops):
    print("== memory-bound kernels at level 0 ([524288, 320] tokens); median ms / effective GB/s (algorithmic bytes)")
    M, C, V, F, L = 524288, 320, 8, 16, 4096
    x = rnd(M, C)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    med, _ = timeit(lambda: ops.group_norm(x, 128, L, g, b, 32, 1e-5, True)); print(f"group_norm 2D     : {med:7.3f} ms  {3 * M * C * 2 / med / 1e6:7.0f} GB/s")
    med, _ = timeit(lambda: ops.group_norm(x, V, F * L, g, b, 32, 1e-6, False)); print(f"group_norm 3D     : {med:7.3f} ms  {3 * M * C * 2 / med / 1e6:7.0f} GB/s")
    med, _ = timeit(lambda: ops.layer_norm(x, g, b, 1e-5)); print(f"layer_norm        : {med:7.3f} ms  {2 * M * C * 2 / med / 1e6:7.0f} GB/s")
    pe1, pe2 = rnd(F, C), rnd(L, C)
    med, _ = timeit(lambda: ops.layer_norm(x, g, b, 1e-5, pe1=pe1, pe1_div=L, pe2=pe2, pe2_div=1, two=True)); print(f"layer_norm 2 outs : {med:7.3f} ms  {3 * M * C * 2 / med / 1e6:7.0f} GB/s")
    u = rnd(M, 8 * C)
    med, _ = timeit(lambda: ops.geglu(u)); print(f"geglu             : {med:7.3f} ms  {M * 12 * C * 2 / med / 1e6:7.0f} GB/s")
    qkv = rnd(M, 3 * C)
    med, _ = timeit(lambda: ops.temporal_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], V, F, L, 8)); print(f"temporal_attn D40   : {med:7.3f} ms  {4 * M * C * 2 / med / 1e6:7.0f} GB/s")
    for (M2, C2, L2) in ((131072, 640, 1024), (32768, 1280, 256)):
        qkv2 = rnd(M2, 3 * C2)
        med, _ = timeit(lambda: ops.temporal_attn(qkv2[:, :C2], qkv2[:, C2:2 * C2], qkv2[:, 2 * C2:], V, F, L2, 8)); print(f"temporal_attn C={C2}: {med:7.3f} ms  {4 * M2 * C2 * 2 / med / 1e6:7.0f} GB/s")
    a, bb = rnd(M, 640), rnd(M, 320)
    med, _ = timeit(lambda: ops.concat(a, bb)); print(f"concat 640+320    : {med:7.3f} ms  {2 * M * 960 * 2 / med / 1e6:7.0f} GB/s")


if __name__ == "__main__":
    which = sys.argv[1:] or ["flash", "gemm", "conv", "misc"]
    torch.manual_seed(0)
    ops = HipOps()
    print(torch.cuda.get_device_name(0))
    for w in which:
        {"flash": bench_flash, "vae": bench_vae, "loop": bench_loop, "graph": bench_graph, "gemm": bench_gemm, "conv": bench_conv, "misc": bench_misc,
         "flashdm": bench_flashdm, "smallm": bench_smallm, "gn": bench_gn, "onewave": bench_onewave, "shortk": bench_shortk, "directepi": bench_directepi, "flashspread": bench_flashspread, "flashdt": bench_flashdt, "flashrank": bench_flashrank, "gemmscale": bench_gemmscale, "gemmcal": bench_gemmcal, "wgrad": bench_wgrad, "attnbwd": bench_attnbwd, "flash16": bench_flash16,
         "flash40": lambda o: bench_flash(o, ((40, 4, 16, 4096, 2),)),
         "flashshort": lambda o: bench_flash(o, ((80, 4, 16, 256, 2), (80, 4, 16, 128, 2), (80, 4, 16, 512, 2), (80, 1, 16, 1024, 2), (40, 4, 16, 256, 2), (40, 4, 16, 64, 2), (160, 4, 16, 256, 2), (160, 4, 16, 64, 2))),
         "flash40s": lambda o: bench_flash(o, ((40, 4, 16, 1024, 2), (40, 4, 16, 256, 2), (40, 1, 16, 4096, 2))),
         "flash80": lambda o: bench_flash(o, ((80, 4, 16, 1024, 2), (80, 8, 32, 1024, 1), (80, 2, 3, 96, 2))),
         "flash160": lambda o: bench_flash(o, ((160, 4, 16, 256, 2), (160, 4, 16, 64, 2))),
         "gemm1": lambda o: ([o.gemm(rnd(32768, 5120), rnd(1280, 5120, scale=0.01)) for _ in range(3)],
                             [o.gemm(rnd(524288, 320), rnd(1280, 320, scale=0.05)) for _ in range(3)],
                             [o.conv3x3(rnd(128 * 64 * 64, 320), 128, 64, 64, rnd(320, 2880, scale=0.02), torch.zeros(320, device="cuda")) for _ in range(3)])}[w](ops)
