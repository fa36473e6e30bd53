#!/usr/bin/env python
"""Benchmark of the MV-VDM denoise step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one MVUNetMotionModel.forward on the CFG-doubled batch exactly as the reference pipeline
issues it (pipeline.py:1008-1020): V = 2 x 4 views, 16 frames, 64x64 latent (512^2 images), bf16,
synthetic seeded weights and inputs (no checkpoints / datasets exist offline).  Inputs are resident in
HBM before the timed region.  For N > 1 the SAME job is sharded over the GPUs (CFG halves x views,
animate3d_amd/parallel.py) => strong scaling; value = steps/s of the whole job.

Rank 0 prints ONE JSON line.  It carries
  roofline     — the dominant kernel (flash attention, head_dim 40, level-0 multi-view attention):
                 algorithmic FLOPs per launch / mean launch duration measured with HIP events on the
                 launch stream inside the timed region, against the dense bf16 MFMA peak;
  cpu_baseline — the CPU oracle (plain-PyTorch fp32 restatement of the reference forward; the reference
                 itself cannot be imported offline) timed on this host on BASELINE config 1.
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

# HBM traffic of one level-0 attention launch (PMC, not measurable from inside the process): see profiles/r1_flash_pmc_traffic.md
METRIC = "UNet denoise-steps/sec, 4view\u00d716frame\u00d7512\u00b2 MV-VDM @1/2/4/8 GPU"     # BASELINE.json, verbatim
TRAFFIC_BYTES_PER_LAUNCH = 2.82e9
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


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
    """Delegates to HipOps; brackets every launch of the dominant kernel with HIP events on the stream
    the kernel is launched on (torch's current stream == the stream handed to the C-ABI)."""

    def __init__(self, ops, head_dim, min_kv):
        self._ops, self._hd, self._min_kv = ops, head_dim, min_kv
        self.events, self.flops, self.enabled = [], [], False

    def __getattr__(self, name):
        return getattr(self._ops, name)

    def flash_attn(self, q, k, v, qmap, kmap, groups, heads, q_len, kv_len, **kw):
        D = q.shape[1] // heads
        if not (self.enabled and D == self._hd and kv_len >= self._min_kv):
            return self._ops.flash_attn(q, k, v, qmap, kmap, groups, heads, q_len, kv_len, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = self._ops.flash_attn(q, k, v, qmap, kmap, groups, heads, q_len, kv_len, **kw)
        e1.record()
        self.events.append((e0, e1))
        self.flops.append(4.0 * groups * q_len * kv_len * heads * D)     # QK^T + PV, 1 MAC = 2 FLOP
        return out


def cpu_baseline(threads):
    """Bounded CPU sample: the oracle (plain-PyTorch fp32 restatement of the reference forward) on
    BASELINE config 1 (1 view x 4 frames x 64x64 latent = 512^2 px, fp32, no CFG; 6.45 TFLOP), one forward timed on
    `threads` host threads (about 15 s).  The config-2 figure is a FLOP-ratio extrapolation."""
    from animate3d_amd.config import UNetConfig
    from animate3d_amd.flops import step_flops
    from oracle import unet_ref as O
    torch.set_num_threads(threads)
    cfg = O.UNetConfig()
    hw = (64, 64)
    ref = O.build_fast(cfg, 1, 4, hw, seed=None)      # constant weights: timing only
    inp = O.synthetic_inputs(cfg, 1, 1, 4, hw, seed=1)
    t0 = time.time()
    ref(**inp)
    dt = time.time() - t0
    f_sample = step_flops(UNetConfig(), 1, 1, 4, *hw)["total"]
    f_cfg1 = step_flops(UNetConfig(), 1, 1, 4, 64, 64)["total"]
    f_cfg2 = step_flops(UNetConfig(), 8, 4, 16, 64, 64)["total"]
    return {"value": 1.0 / dt, "unit": "denoise-steps/s", "cores": threads, "kind": "port",
            "sample": f"1 forward of BASELINE config 1: 1 view x 4 frames x 64x64 latent, fp32, no CFG ({f_sample / 1e12:.2f} TFLOP/step); "
                      f"CPU oracle = plain-PyTorch restatement of the reference forward "
                      f"(the reference itself needs diffusers/xformers, absent offline); host has {os.cpu_count()} logical CPUs",
            "seconds_per_step": dt, "tflops": f_sample / dt / 1e12,
            "config1_equivalent_steps_per_s": (1.0 / dt) * f_sample / f_cfg1,
            "config2_equivalent_steps_per_s": (1.0 / dt) * f_sample / f_cfg2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (default min(host CPUs, 16))")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from animate3d_amd.config import UNetConfig
    from animate3d_amd.hip_ops import HipOps
    from animate3d_amd.unet import MVUNetMotionModel

    cfg = UNetConfig()
    n, F, hw = args.views, args.frames, (args.latent, args.latent)
    V = 2 * n
    S0 = n * hw[0] * hw[1]
    ops = TimedOps(HipOps(dev), head_dim=cfg.block_out_channels[0] // cfg.num_attention_heads, min_kv=S0 if world == 1 else S0)
    model = MVUNetMotionModel(cfg, ops=ops, num_views=n, device=dev)
    model.init_synthetic(seed=0)
    model = model.to(torch.bfloat16).eval()
    if world > 1:
        from animate3d_amd.parallel import shard_unet
        shard_unet(model)
    inp = make_inputs(cfg, V, n, F, hw, dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model(**inp)
    ops.enabled = True
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = model(**inp).sample
    sync()
    dt = time.perf_counter() - t0
    ops.enabled = False
    assert torch.isfinite(y).all()
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    ms_per_step = dt / args.steps * 1e3
    value = args.steps / dt

    # dominant kernel, measured live with HIP events on the launch stream
    durs = [a.elapsed_time(b) * 1e-3 for a, b in ops.events]
    if durs:
        mean_dur = sum(durs) / len(durs)
        flops = sum(ops.flops) / len(ops.flops)
        achieved = flops / mean_dur / 1e12
        roofline = {"bound": "mfma", "kernel": "flash_attn_kernel<40,64> (level-0 multi-view / first-frame attention)",
                    "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                    "traffic": TRAFFIC_BYTES_PER_LAUNCH if world == 1 else None, "traffic_unit": "HBM bytes per launch",
                    "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2: gfx950 counts 128-B requests as 64 B) and WRITE_SIZE, separate "
                                      "passes, same launch shape (profiles/r1_flash_pmc_traffic.md); algorithmic bytes 1.34e9",
                    "launches_per_step": len(durs) // args.steps, "mean_launch_ms": mean_dur * 1e3,
                    "flop_per_launch": flops, "share_of_step_time": sum(durs) / dt}
    else:
        roofline = None

    if rank == 0:
        from animate3d_amd.flops import step_flops
        work = step_flops(cfg, V, n, F, *hw)["total"]      # 394.78 TFLOP at config 2
        line = {
            "metric": METRIC, "value": value, "unit": "denoise-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: {n} views x {F} frames x {hw[0] * 8}^2 px ({hw[0]}x{hw[1]} latent), CFG-doubled "
                                   f"batch V={V} videos, one MVUNetMotionModel.forward per step, SD1.5 MV-VDM UNet 1.53 B params, "
                                   "seeded synthetic weights",
                       "parallelism": "single GPU" if world == 1 else f"cfg{model.parallel.cfg_shards} x views{model.parallel.view_shards} (K|V all-gather over RCCL)"},
            "config2_25_ddim_steps_seconds": 25.0 * ms_per_step / 1e3,
            "flop_per_step": work, "whole_step_tflops": work * value / 1e12,
            "whole_step_mfma_frac": work * value / 1e12 / (PEAK_BF16_TFLOPS * world),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_threads or min(os.cpu_count() or 1, 16))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
