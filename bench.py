#!/usr/bin/env python
"""Benchmark of the MV-VDM denoise step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W [--layout cfg,views,frames] [--gather-kv]
    python bench.py --rank-shape 2,4,1                     # ONE GPU: the compute leg of rank 0 of an 8-GPU layout (collectives replaced by local
                                                           # copies of the right size; link time modelled, not measured); the line's metric says so

One "step" = one MVUNetMotionModel.forward on the CFG-doubled batch exactly as the reference pipeline
issues it (pipeline.py:1008-1020).  Default workload = BASELINE config 2 (the configuration the metric is quoted on):
V = 2 x 4 views, 16 frames, 64x64 latent (512^2 images), bf16, synthetic seeded weights and inputs (no checkpoints /
datasets exist offline).  ``--config 4`` (8 views x 32 frames x 64x64 latent) and ``--config 5`` (the 4D-SDS call:
4 views x 16 frames x 32x32 latent) run the other single-GPU-sized BASELINE configurations through the same path.
Inputs are resident in HBM before the timed region.  For N > 1 the SAME job is sharded over the GPUs (CFG halves x views x
frames, animate3d_amd/parallel.py) => strong scaling; value = steps/s of the whole job.

Rank 0 prints ONE JSON line.  It carries
  roofline     — the dominant kernel as rocprofv3 names it (flash_attn_dm_kernel<5, 2> in both storage types:
                 level-0 multi-view / first-frame attention, head_dim 40): algorithmic FLOPs per launch / mean launch duration measured with HIP events on the
                 launch stream inside the timed region, against the dense bf16 MFMA peak; ``ceilings`` adds the second
                 ceiling that binds at head_dim 40 — the v_exp issue rate measured with tools/ubench_exp.hip — and ``traffic``
                 the HBM bytes per launch from the rocprofv3 PMC pass committed under profiles/ (read from
                 profiles/r6_flash_pmc_traffic.json (a pass over the final tree of round 6), else older rounds'; null when no file describes this kernel and launch shape);
  groups       — per kernel family (attention by head dim, GEMM, fused GEGLU GEMM, 3x3 conv, norms, ...): ms per step and
                 achieved TFLOP/s or GB/s, from one extra instrumented forward OUTSIDE the timed region;
  cpu_baseline — the CPU oracle (plain-PyTorch fp32 restatement of the reference forward; the reference itself cannot be
                 imported offline) timed on this host on BASELINE config 1: seeded weights, 1 warm-up + 3 timed forwards
                 (fewer only past a 150-s budget), thread count chosen by a 2-second matmul probe over the CPUs this process may use.
  box          — calibration kernels outside the timed region (8192^3 GEMM, the level-0 attention launch on zero-filled and on random
                 inputs, a 336 MB LayerNorm, the shader clock if rocm-smi answers): compute-bound kernels run at a power-limited clock
                 that differs by several per cent between boxes, so round-over-round deltas are quoted against these.
``--dtype`` selects the storage type of the kernels; the default is the dtype BASELINE.json states for the configuration (2: bf16, 4 and
5: fp16) and the line's ``dtype`` reports what ran.  ``--graph`` adds the same steps replayed from a HIP graph (capture_graph).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "UNet denoise-steps/sec, 4view×16frame×512² MV-VDM @1/2/4/8 GPU"     # BASELINE.json, verbatim
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0            # HBM3E spec; ~6300 GB/s achievable (MI355X_MICROARCH.md)
# v_exp_f32 issue rate next to the kernel's own MFMA / v_cvt_pk mix, two waves per SIMD (tools/ubench_exp.hip,
# profiles/r2_ubench_exp.log): 9.4e12 exp/s chip-wide without the v_max3 chain of the round-2 kernel (8.5e12 with it); every score
# costs one exp and 4 * head_dim MFMA FLOPs
EXP_PER_S_IN_MIX = {"bf16": 9.4e12, "fp16": 9.4e12}
# level-0 attention kernel per storage type, and the share of its issued MFMA work that is useful (QK^T contraction 40 -> 48;
# O^T rows 41 -> 48 through 16x16x32); fp16 storage runs the same kernel with a sampled softmax offset (csrc/flash_attn_dm.hip, DM_BIAS)
DOMINANT_KERNEL = {"bf16": "flash_attn_dm_kernel<5, 2>", "fp16": "flash_attn_dm_kernel<5, 2>"}
USEFUL_MFMA_SHARE = {"bf16": 640.0 / 768.0, "fp16": 640.0 / 768.0}
CONFIG_DTYPE = {2: "bf16", 4: "fp16", 5: "fp16"}      # BASELINE.json

CONFIGS = {     # BASELINE.json configs that fit one GPU: (views, frames, latent, label)
    2: (4, 16, 64, "BASELINE config 2: 4 views x 16 frames x 512^2 px (64x64 latent)"),
    4: (8, 32, 64, "BASELINE config 4 on ONE GPU: 8 views x 32 frames x 512^2 px (64x64 latent)"),
    5: (4, 16, 32, "BASELINE config 5 (UNet call of one 4D-SDS step): 4 views x 16 frames x 256^2 px (32x32 latent)"),
}


def make_inputs(cfg, V, n, F, hw, device, seed=1):
    from animate3d_amd.embeddings import get_camera
    g = torch.Generator(device="cpu").manual_seed(seed)
    h, w = hw
    sample = torch.randn(V, cfg.in_channels, F, h, w, generator=g)
    sample[:, :, 0] *= 0.18215                                   # clean first frame (pipeline.py:951-953)
    ehs = torch.randn(V, 77, cfg.cross_attention_dim, generator=g)
    img = torch.randn(V, cfg.ip_image_embed_dim, generator=g)
    img[: V // 2] = 0.0                                          # (uncond, text) order, pipeline.py:937
    cam = get_camera(n).repeat(V // n, 1)
    return dict(sample=sample.to(device), timestep=torch.tensor(501, device=device), encoder_hidden_states=ehs.to(device),
                added_cond_kwargs={"image_embeds": img.to(device)}, camera=cam.to(device), num_views=n)


class TimedOps:
    """Delegates to HipOps.  ``enabled``: brackets every launch of the dominant kernel with HIP events on the stream the
    kernel is launched on (torch's current stream == the stream handed to the C-ABI).  ``profile``: brackets EVERY op and
    books duration + algorithmic work per kernel family (used for one extra forward outside the timed region)."""

    def __init__(self, ops, head_dim, min_kv):
        self._ops, self._hd, self._min_kv = ops, head_dim, min_kv
        self.events, self.flops, self.enabled = [], [], False
        self.profile, self.records = False, []

    _PASS_THROUGH = ("reserved_cus", "split_k")      # per-call launch options that animate3d_amd.parallel sets on the op set it is handed

    def __setattr__(self, name, value):
        if name in TimedOps._PASS_THROUGH:
            setattr(self._ops, name, value)
        else:
            object.__setattr__(self, name, value)

    def __getattr__(self, name):
        fn = getattr(self._ops, name)
        if not (self.profile and callable(fn)) or name.startswith("_") or name in ("empty", "interleave_geglu", "split_cols"):
            return fn

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.records.append((self._family(name, a, k), e0, e1, self._work(name, a, k, out), self._shape(name, a, k)))
            return out
        return timed

    @staticmethod
    def _shape(name, a, k):
        if name in ("gemm", "gemm_geglu"):
            return f"{name} M={a[0].shape[0]} N={a[1].shape[0]} K={a[0].shape[1]}" + (" +res" if k.get("residual") is not None else "") + (" +rowbias" if k.get("rowbias") is not None else "")
        if name == "gemm2":
            return f"gemm M={a[0].shape[0]} N={a[2].shape[0]} K={a[0].shape[1]}+{a[1].shape[1]} (two-source A)"
        if name == "group_norm2":
            return f"group_norm ({a[0].shape[0]}, {a[0].shape[1]}+{a[1].shape[1]}) (two-source)"
        if name == "conv3x3":
            return f"conv3x3 B={a[1]} {a[2]}x{a[3]} Cin={a[0].shape[1]} Cout={a[4].shape[0]}" + (f" stride={k['stride']}" if k.get("stride", 1) != 1 else "") + (" up2x" if k.get("up2x") else "") + (" +res" if k.get("residual") is not None else "")
        if name in ("group_norm", "layer_norm", "concat", "temporal_attn"):
            return f"{name} {tuple(a[0].shape)}"
        return name

    @staticmethod
    def _family(name, a, k):
        if name in ("gemm", "gemm2"):
            return "gemm"
        if name == "group_norm2":
            return "group_norm"
        if name == "conv3x3":
            return "conv3x3 (nearest-2x up)" if k.get("up2x") else "conv3x3"
        return name

    @staticmethod
    def _work(name, a, k, out):
        """(flops, bytes) of one call: algorithmic — operands read once, result written once."""
        by = lambda *ts: float(sum(t.numel() * t.element_size() for t in ts if torch.is_tensor(t)))
        if name in ("gemm", "gemm_geglu", "gemm_f32out"):
            x, w = a[0], a[1]
            return 2.0 * x.shape[0] * w.shape[0] * x.shape[1], by(x, w, out, k.get("residual"))
        if name == "gemm2":
            xa, xb, w = a[0], a[1], a[2]
            if out is None:
                return 0.0, 0.0
            return 2.0 * xa.shape[0] * w.shape[0] * (xa.shape[1] + xb.shape[1]), by(xa, xb, w, out)
        if name == "group_norm2":
            return 0.0, 3.0 * by(a[0], a[1])
        if name == "conv3x3":
            x, B, H, W, w = a[:5]
            y = out[0]
            return 2.0 * y.shape[0] * w.shape[0] * w.shape[1], by(x, w, y, k.get("residual"))
        if name == "temporal_attn":
            q, kk, v, videos, frames, L, heads = a[:7]
            return 4.0 * frames * q.shape[0] * q.shape[1], by(q, kk, v, out)
        if name in ("group_norm", "group_norm_apply"):
            return 0.0, 3.0 * by(a[0])            # statistics pass + apply pass read, one write
        if name == "layer_norm":
            return 0.0, by(a[0]) * (3.0 if k.get("two") else 2.0)
        return 0.0, by(*a) + (by(out) if torch.is_tensor(out) else 0.0)

    def flash_attn(self, q, k, v, qmap, kmap, groups, heads, q_len, kv_len, **kw):
        D = q.shape[1] // heads
        work = 4.0 * groups * q_len * kv_len * heads * D             # QK^T + PV, 1 MAC = 2 FLOP
        if self.profile:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self._ops.flash_attn(q, k, v, qmap, kmap, groups, heads, q_len, kv_len, **kw)
            e1.record()
            fam = f"flash_attn D={D}" + (" (cross: text / IP tokens)" if kv_len <= 128 else "")
            self.records.append((fam, e0, e1, (work, 0.0), f"{fam} G={groups} Sq={q_len} Skv={kv_len}"))
            return out
        if not (self.enabled and D == self._hd and kv_len >= self._min_kv):
            return self._ops.flash_attn(q, k, v, qmap, kmap, groups, heads, q_len, kv_len, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = self._ops.flash_attn(q, k, v, qmap, kmap, groups, heads, q_len, kv_len, **kw)
        e1.record()
        self.events.append((e0, e1))
        self.flops.append(work)
        return out

    def shape_summary(self, top=90):
        agg = {}
        for _, e0, e1, (fl, by), shp in self.records:
            d = agg.setdefault(shp, [0, 0.0, 0.0, 0.0])
            d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += fl; d[3] += by
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]
        return [{"op": k, "launches": c, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1) if fl else None, "gbs": round(by / ms / 1e6) if by else None}
                for k, (c, ms, fl, by) in rows]

    def group_summary(self):
        agg = {}
        for fam, e0, e1, (fl, by), _ in self.records:
            d = agg.setdefault(fam, [0, 0.0, 0.0, 0.0])
            d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += fl; d[3] += by
        out = {}
        for fam, (cnt, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            rec = {"launches": cnt, "ms": round(ms, 3)}
            if fl > 0:
                rec["tflops"] = round(fl / ms / 1e9, 1)
            if by > 0 and fl == 0:
                rec["gbs"] = round(by / ms / 1e6, 0)
            out[fam] = rec
        return out


def _usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                            # cgroup v2 quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _pick_threads(limit):
    """Fastest thread count for fp32 GEMM on this host among a few candidates (2-second probe): more threads than the box can
    actually run is slower, and os.cpu_count() over-reports inside containers."""
    cands = sorted({t for t in (8, 16, 32, 64, 128, 256) if t <= limit} | {min(limit, 256)})
    a, b = torch.randn(2048, 2048), torch.randn(2048, 2048)
    best, best_t, seen = cands[0], float("inf"), {}
    for t in cands:
        torch.set_num_threads(t)
        a @ b
        t0 = time.perf_counter()
        a @ b
        a @ b
        dt = (time.perf_counter() - t0) / 2
        seen[t] = round(2 * 2048 ** 3 / dt / 1e9, 1)
        if dt < best_t * 0.95:
            best, best_t = t, dt
        if dt > 3 * best_t:
            break
    return best, seen


def cpu_baseline(threads, budget_s=150.0):
    """Bounded CPU sample: the oracle (plain-PyTorch fp32 restatement of the reference forward) on BASELINE config 1
    (1 view x 4 frames x 64x64 latent = 512^2 px, fp32, no CFG; 6.45 TFLOP): seeded weights, 1 warm-up + up to 3 timed
    forwards, stopping when the time budget is used.  The config-2 figure is a FLOP-ratio extrapolation."""
    from animate3d_amd.config import UNetConfig
    from animate3d_amd.flops import step_flops
    from oracle import unet_ref as O
    usable = _usable_cpus()
    probe = None
    if not threads:
        threads, probe = _pick_threads(usable)
    torch.set_num_threads(threads)
    cfg = O.UNetConfig()
    hw = (64, 64)
    ref = O.build_fast(cfg, 1, 4, hw, seed=0)
    inp = O.synthetic_inputs(cfg, 1, 1, 4, hw, seed=1)
    t_all = time.time()
    t0 = time.time(); ref(**inp); warm = time.time() - t0
    runs = []
    while len(runs) < 3 and (not runs or (time.time() - t_all) + min(runs) < budget_s):
        t0 = time.time(); ref(**inp); runs.append(time.time() - t0)
    dt = min(runs)
    f_cfg1 = step_flops(UNetConfig(), 1, 1, 4, 64, 64)["total"]
    f_cfg2 = step_flops(UNetConfig(), 8, 4, 16, 64, 64)["total"]
    return {"value": 1.0 / dt, "unit": "denoise-steps/s", "cores": threads, "kind": "port",
            "sample": f"BASELINE config 1: 1 view x 4 frames x 64x64 latent, fp32, no CFG ({f_cfg1 / 1e12:.2f} TFLOP/step); seeded weights, "
                      f"1 warm-up ({warm:.1f} s) + {len(runs)} timed forwards (best of {[round(r, 1) for r in runs]} s); CPU oracle = "
                      f"plain-PyTorch restatement of the reference forward (the reference itself needs diffusers/xformers, absent "
                      f"offline); host reports {os.cpu_count()} logical CPUs, {usable} usable by this process",
            "thread_probe_gflops": probe, "seconds_per_step": dt, "tflops": f_cfg1 / dt / 1e12,
            "config2_equivalent_steps_per_s": (1.0 / dt) * f_cfg1 / f_cfg2}


def box_calibration(ops, dev):
    """Fixed kernels outside the timed region that tell this box from the next one (compute-bound kernels run at a power-limited clock that
    differs by several per cent between boxes and between hours on one box): the 8192^3 GEMM on uniform random operands, the level-0
    attention launch of config 2 on ZERO-filled inputs (no data-dependent switching power: the clock ceiling), the same launch on
    random inputs, and a 336 MB LayerNorm (HBM).  Round-over-round deltas in DESIGN.md are quoted against these."""
    from animate3d_amd.hip_ops import RowMap
    dt = ops.act_dtype

    def med(fn, reps, warm=2):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    g = torch.Generator(device=dev).manual_seed(7)
    a = (torch.rand(8192, 8192, device=dev, generator=g) - 0.5).to(dt)
    b = (torch.rand(8192, 8192, device=dev, generator=g) - 0.5).to(dt)
    t_gemm = med(lambda: ops.gemm(a, b), 9)
    del a, b
    n, F, L, bb, C, heads = 4, 16, 4096, 2, 320, 8
    qm = RowMap(F, n * F * L, L, L, F * L)
    S, G = n * L, bb * F
    qkv = torch.zeros(bb * n * F * L, 3 * C, device=dev, dtype=dt)
    attn = lambda: ops.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], qm, qm, G, heads, S, S)
    t_zero = med(attn, 3, warm=1)
    qkv.copy_(torch.randn(qkv.shape, device=dev, generator=g))
    t_rand = med(attn, 3, warm=1)
    x = qkv[:, :C].contiguous()
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    t_ln = med(lambda: ops.layer_norm(x, gam, bet, 1e-5), 9)
    box = {"gemm8192_tflops": round(2.0 * 8192 ** 3 / t_gemm / 1e9, 1), "attn_zero_ms": round(t_zero, 3), "attn_randn_ms": round(t_rand, 3),
           "layer_norm_336MB_gbs": round(2.0 * x.numel() * 2 / t_ln / 1e6), "device": torch.cuda.get_device_name(dev)}
    try:       # shader clock right after the calibration kernels, if the SMI tool answers in time (informational)
        import re
        import subprocess
        txt = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        m = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)\s*Mhz", txt, re.I)
        box["sclk_mhz"] = int(m.group(1)) if m else None
    except Exception:
        box["sclk_mhz"] = None
    return box


def _pmc_traffic(S0, groups, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass, if it matches this kernel and launch shape."""
    for name in ("r6_flash_pmc_traffic.json", "r4_flash_pmc_traffic.json", "r3_flash_pmc_traffic.json"):      # newest pass first (round 6: re-run on the final tree)
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
            if rec.get("kernel") == kernel and rec.get("kv_len") == S0 and rec.get("groups") == groups:
                return rec
        except Exception:
            pass
    return None


XGMI_LINK_GBS = 153.0            # per link and direction (MI355X_MICROARCH.md); a gather among S peers uses S - 1 links of a GPU concurrently


def _link_ms(plan, steps):
    """Modelled link time of the gathers a rank of this layout would receive per step: bytes / ((peers of the gathering group) x one xGMI
    link).  View gathers use view_shards - 1 links, frame gathers frame_shards - 1; the plan counts bytes only, so the split is by axis
    share when both are cut (views move ~60 % of the bytes at config 2: DESIGN.md section 5)."""
    by = plan.gather_bytes / max(1, steps)
    peers = max(1, max(plan.view_shards, plan.frame_shards) - 1)
    return by / (peers * XGMI_LINK_GBS * 1e9) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configuration (default 2: the one the metric is quoted on)")
    ap.add_argument("--views", type=int, default=0)
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--latent", type=int, default=0)
    ap.add_argument("--layout", type=str, default="", help="multi-GPU: cfg,views,frames shard counts (default: CFG halves, then views, then frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-groups", action="store_true", help="skip the instrumented forward behind the 'groups' object")
    ap.add_argument("--dtype", choices=("bf16", "fp16"), default=None, help="kernel storage type (default: what BASELINE.json states for the configuration)")
    ap.add_argument("--graph", action="store_true", help="also time the same steps replayed from a HIP graph (capture_graph)")
    ap.add_argument("--shapes", action="store_true", help="print the instrumented forward per op shape on stderr (top 90 by time)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (default: picked by a short GEMM probe)")
    ap.add_argument("--per-layer-conditioning", action="store_true", help="A/B: project the time embedding / text / IP tokens once per layer (54 small GEMMs, "
                    "rounds 1-5) instead of stacked (3 GEMMs, MVUNetMotionModel.stack_conditioning)")
    ap.add_argument("--no-box", action="store_true", help="skip the box calibration kernels behind the 'box' object")
    ap.add_argument("--gather-kv", action="store_true", help="sharded / --rank-shape runs: all-gather the projected K|V (2C wide) instead of the attention's "
                    "input tokens (C wide, K|V projected locally for all views): ShardPlan.gather_tokens = False")
    ap.add_argument("--rank-shape", type=str, default="", help="single process, no process group: time the COMPUTE leg of rank 0 of the multi-GPU layout "
                    "cfg,views,frames (its local rows, K|V projections over the gathered token count, q_len != kv_len attention, split-K off, CU "
                    "reservation while a gather would be in flight), every collective replaced by a local copy of the right size; the line's metric says so")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher — one rank per GPU under torch.distributed.run on a free local port; the
        # ranks are this same script (WORLD_SIZE set), rank 0 prints the one JSON line on the inherited stdout
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # --standalone: the c10d rendezvous picks its own free port (no bind-then-close probe that another process can win on a busy box)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or without a launcher)")
    # A3D_BENCH_SHARE_GPU=1 (testing the multi-rank code path on a one-GPU box): all ranks on device 0, collectives over gloo
    share = os.environ.get("A3D_BENCH_SHARE_GPU") == "1"
    dev_id = 0 if share else local_rank
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from animate3d_amd.config import UNetConfig
    from animate3d_amd.hip_ops import HipOps
    from animate3d_amd.unet import MVUNetMotionModel

    cfg = UNetConfig()
    n0, F0, lat0, label = CONFIGS[args.config]
    n, F, lat = args.views or n0, args.frames or F0, args.latent or lat0
    hw = (lat, lat)
    if (n, F, lat) != (n0, F0, lat0):
        label = f"custom: {n} views x {F} frames x {lat * 8}^2 px ({lat}x{lat} latent)"
    V = 2 * n
    S0 = n * hw[0] * hw[1]
    dtype_name = args.dtype or CONFIG_DTYPE[args.config]
    torch_dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    ops = TimedOps(HipOps(dev, act_dtype=torch_dtype), head_dim=cfg.block_out_channels[0] // cfg.num_attention_heads, min_kv=S0)
    model = MVUNetMotionModel(cfg, ops=ops, num_views=n, device=dev)
    model.init_synthetic(seed=0)
    model = model.to(torch_dtype).eval()
    model.stack_conditioning = not args.per_layer_conditioning
    assert model.ops.act_dtype == torch_dtype
    dom_kernel = DOMINANT_KERNEL[dtype_name]
    par = rpar = None
    if args.rank_shape:
        if world > 1:
            raise SystemExit("--rank-shape emulates one rank in a single process: use it with --gpus 1")
        from animate3d_amd.parallel import rank_shape_unet
        rpar = rank_shape_unet(model, tuple(int(v) for v in args.rank_shape.split(",")))
        rpar.configure(V // n, n, F)
        rpar.gather_tokens = not args.gather_kv
    if world > 1:
        from animate3d_amd.parallel import shard_unet
        layout = tuple(int(v) for v in args.layout.split(",")) if args.layout else None
        par = shard_unet(model, layout=layout, shape=(V // n, n, F))
        par.gather_tokens = not args.gather_kv
    inp = make_inputs(cfg, V, n, F, hw, dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(k):
        sync()
        t0 = time.perf_counter()
        for _ in range(k):
            out = model(**inp).sample
        sync()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = tmax.item()
        return dt_, out

    for _ in range(args.warmup):
        model(**inp)
    for p_ in (par, rpar):
        if p_ is not None:
            p_.gather_bytes = p_.collectives = 0
    ops.enabled = True
    dt, y = timed_steps(args.steps)
    ops.enabled = False
    assert torch.isfinite(y).all()
    ms_per_step = dt / args.steps * 1e3
    value = args.steps / dt

    # dominant kernel, measured live with HIP events on the launch stream
    durs = [a.elapsed_time(b) * 1e-3 for a, b in ops.events]
    roofline = None
    if durs:
        mean_dur = sum(durs) / len(durs)
        flops = sum(ops.flops) / len(ops.flops)
        achieved = flops / mean_dur / 1e12
        D = ops._hd
        exp_ceiling = EXP_PER_S_IN_MIX[dtype_name] * 4.0 * D / 1e12
        pmc = _pmc_traffic(S0, (V // n) * F, dom_kernel) if world == 1 and dtype_name == "bf16" else None      # the committed PMC pass ran the bf16 build
        roofline = {"bound": "mfma", "kernel": dom_kernel + " (level-0 multi-view / first-frame attention, head_dim 40)",
                    "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                    "ceilings": {"mfma_dense_bf16_tflops": PEAK_BF16_TFLOPS,
                                 "mfma_after_padding_tflops": round(PEAK_BF16_TFLOPS * USEFUL_MFMA_SHARE[dtype_name], 1),
                                 "exp_issue_tflops_equivalent": round(exp_ceiling, 1), "frac_of_exp_ceiling": achieved / exp_ceiling,
                                 "note": "head_dim 40: one v_exp_f32 per 160 MFMA FLOPs; padding of the issued MFMA work: QK^T contraction 40->48, O^T rows "
                                         "41->48 (16x16x32 MFMA); the exp ceiling is the v_exp rate measured next to the "
                                         "kernel's MFMA / cvt mix (tools/ubench_exp.hip); the kernel is clock-limited by power: zero inputs run the shipped variant 13-14 % faster, "
                                         "removing its per-tile barrier changes nothing on random data, and a one-wave-per-SIMD re-instantiation (round 4) lands on "
                                         "the same 10.8 ms (profiles/README.md).  Score spread: the seeded synthetic weights give a nearly flat softmax (score sd ~0.3); "
                                         "the same launch on q, k with score sd 1 / 3 / 6 takes +1.5 / +3.4 / +3.3 % in bf16 storage (10.96 -> 11.13 / 11.33 / 11.32 ms, no "
                                         "exact re-runs: tools/microbench.py flashspread, profiles/r5_flash_score_spread.log)"},
                    "traffic": (pmc or {}).get("hbm_bytes_per_launch"), "traffic_unit": "HBM bytes per launch",
                    "traffic_source": (pmc or {}).get("source", "no committed PMC pass for this launch shape"),
                    "algorithmic_bytes_per_launch": 4.0 * (V // n) * F * S0 * D * 8 * 2,      # Q, K, V read + O written once, bf16
                    "launches_per_step": len(durs) // args.steps, "mean_launch_ms": mean_dur * 1e3,
                    "flop_per_launch": flops, "share_of_step_time": sum(durs) / dt}

    comm = None
    if par is not None:      # exposed communication = step time with the data-path collectives replaced by local copies
        comm = {"layout": {"cfg": par.cfg_shards, "views": par.view_shards, "frames": par.frame_shards},
                "received_bytes_per_rank_per_step": par.gather_bytes / args.steps, "collectives_per_step": par.collectives / args.steps}
        import torch.distributed as dist_
        real = (dist_.all_gather_into_tensor, dist_.broadcast, dist_.all_reduce)

        class _Done:
            def wait(self):
                return True

        def fake_gather(out, t, group=None, async_op=False):
            if group is None:                       # the final output gather stays real
                return real[0](out, t, group=group, async_op=async_op)
            out.view(-1, *t.shape)[:] = t
            return _Done() if async_op else None
        dist_.all_gather_into_tensor = fake_gather
        dist_.broadcast = lambda buf, src, group=None: None
        dist_.all_reduce = lambda t, op=None, group=None: (real[2](t, op=op, group=group) if group is None else None)
        try:
            dry, _ = timed_steps(max(1, min(args.steps, 3)))
        finally:
            dist_.all_gather_into_tensor, dist_.broadcast, dist_.all_reduce = real
        comm["ms_per_step_without_collectives"] = dry / max(1, min(args.steps, 3)) * 1e3
        comm["exposed_communication_ms_per_step"] = ms_per_step - comm["ms_per_step_without_collectives"]

    graph_ms = None
    if args.graph and world == 1:
        gstep = model.capture_graph(**inp)
        for _ in range(max(1, args.warmup)):
            gstep(**inp)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            gstep(**inp)
        sync()
        graph_ms = (time.perf_counter() - t0) / args.steps * 1e3

    groups = None
    if not args.no_groups:
        ops.profile = True
        model(**inp)
        torch.cuda.synchronize()
        ops.profile = False
        groups = ops.group_summary()
        if args.shapes and rank == 0:
            for r in ops.shape_summary():
                print(f"[shape] {r['ms']:8.3f} ms  x{r['launches']:3d}  {r['tflops'] or '':>7} TF/s  {r['gbs'] or '':>6} GB/s  {r['op']}", file=sys.stderr)

    if rank == 0:
        from animate3d_amd.flops import step_flops
        work = step_flops(cfg, V, n, F, *hw)["total"]      # 394.78 TFLOP at config 2
        line = {
            "metric": METRIC, "value": value, "unit": "denoise-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": f"{label}, CFG-doubled batch V={V} videos, one MVUNetMotionModel.forward per step, SD1.5 MV-VDM UNet "
                                   "1.53 B params, seeded synthetic weights",
                       "parallelism": "single GPU" if world == 1 else
                       f"cfg{par.cfg_shards} x views{par.view_shards} x frames{par.frame_shards} (token / K|V all-gathers over RCCL)"},
            "flop_per_step": work, "whole_step_tflops": work * value / 1e12,
            "whole_step_mfma_frac": work * value / 1e12 / (PEAK_BF16_TFLOPS * world),
            "roofline": roofline,
        }
        if rpar is not None:
            # NOT the BASELINE metric: one rank's compute time of an N-GPU layout measured on one GPU (link time excluded)
            P = rpar.world
            line["metric"] = (f"compute leg of rank 0 of layout cfg{rpar.cfg_shards} x views{rpar.view_shards} x frames{rpar.frame_shards} "
                              f"({P} GPUs), ms per denoise step, collectives replaced by local copies (single-GPU emulation; no link time)")
            line["value"], line["unit"], line["higher_is_better"] = ms_per_step, "ms", False
            line["config"]["parallelism"] = f"rank 0 of {P}: cfg{rpar.cfg_shards} x views{rpar.view_shards} x frames{rpar.frame_shards}, emulated in one process"
            line["flop_per_step"] = work / P
            line["whole_step_tflops"] = work / P / (ms_per_step * 1e-3) / 1e12
            line["whole_step_mfma_frac"] = line["whole_step_tflops"] / PEAK_BF16_TFLOPS
            line["rank_shape"] = {"layout": [rpar.cfg_shards, rpar.view_shards, rpar.frame_shards], "ranks": P,
                                  "algorithmic_flop_per_rank": work / P,
                                  "would_receive_bytes_per_step": rpar.gather_bytes / args.steps, "collectives_per_step": rpar.collectives / args.steps,
                                  "modelled_link_ms": _link_ms(rpar, args.steps),
                                  "reserved_cus_during_gathers": rpar.reserve_cus,
                                  "gathers": "attention input tokens (K|V projected locally for all views)" if rpar.gather_tokens else "projected K|V"}
        if rpar is None and args.config == 2 and (n, F, lat) == (n0, F0, lat0):
            line["config2_25_ddim_steps_seconds"] = 25.0 * ms_per_step / 1e3
        if graph_ms is not None:
            line["hip_graph_replay"] = {"ms_per_step": graph_ms, "eager_ms_per_step": ms_per_step,
                                        "note": "the same forward captured once (MVUNetMotionModel.capture_graph) and replayed; inputs copied into the static buffers per step"}
        if groups is not None:
            line["groups"] = groups
        if comm is not None:
            line["communication"] = comm
        if world == 1 and not args.no_box:
            line["box"] = box_calibration(ops._ops, dev)
        if world == 1 and not args.no_cpu_baseline and rpar is None:
            line["cpu_baseline"] = cpu_baseline(args.cpu_threads)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
