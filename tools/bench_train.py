"""Training-step timing at the shape of configs/training/train.yaml (batch 1 x 4 views x 16 frames x 256^2 px = 32x32 latent, no CFG):
one ``animate3d_amd.train.training_step`` = UNet forward + backward + clip + AdamW on the HIP kernels.  Prints one JSON line.
(The driver's bench contract is bench.py's inference metric; this is the f4 row's own number.)"""
import argparse
import json
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--checkpoint", action="store_true", help="unet.enable_gradient_checkpointing() (train.py:381-382): one checkpoint per layer")
    args = ap.parse_args()
    from animate3d_amd.config import UNetConfig
    from animate3d_amd.denoise import ddim_schedule
    from animate3d_amd.flops import step_flops
    from animate3d_amd.train import FlatAdamW, select_trainable, training_step
    from animate3d_amd.unet import MVUNetMotionModel
    n, Fr, lat = args.views, args.frames, args.latent
    cfg = UNetConfig()
    model = MVUNetMotionModel(cfg, num_views=n, device="cuda").init_synthetic(seed=0)
    params = select_trainable(model)
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    model.enable_training(compute_dtype=dt)
    if args.checkpoint:
        model.enable_gradient_checkpointing()
    opt = FlatAdamW(params, model.ops, lr=1e-4, loss_scale=None if dt == torch.bfloat16 else 65536.0)
    g = torch.Generator(device="cuda").manual_seed(0)
    latents = torch.randn(1, n, 4, Fr, lat, lat, generator=g, device="cuda") * 0.5
    text = torch.randn(1, 77, 768, generator=g, device="cuda")
    from animate3d_amd.embeddings import get_camera
    cams = get_camera(n).cuda()
    img = torch.randn(n, 1024, generator=g, device="cuda")
    ac = ddim_schedule(25)[1]
    step = lambda: training_step(model, opt, latents, text, cams, img, alphas_cumprod=ac, num_views=n, generator=g)
    for _ in range(args.warmup):
        info = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    # forward-only time of the same call (no_grad: the inference path) for the backward / forward ratio
    x = torch.randn(n, 4, Fr, lat, lat, generator=g, device="cuda")
    t = torch.full((n,), 500, device="cuda")
    fwd = lambda: model(x, t, encoder_hidden_states=text.expand(n, 77, 768), camera=cams, num_views=n, added_cond_kwargs={"image_embeds": img})
    with torch.no_grad():
        fwd(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fwd()
        torch.cuda.synchronize()
    ms_fwd = (time.perf_counter() - t0) / args.steps * 1e3
    f_fwd = step_flops(cfg, n, n, Fr, lat, lat)["total"]
    print(json.dumps({"metric": "training steps/s (train.yaml shape, 1 GPU)", "value": 1e3 / ms, "ms_per_step": ms, "forward_only_ms": ms_fwd,
                      "dtype": args.dtype, "gradient_checkpointing": bool(args.checkpoint), "trainable_params": opt.numel, "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
                      "forward_tflop": f_fwd / 1e12, "approx_step_tflops": 3.0 * f_fwd / (ms * 1e-3) / 1e12,
                      "loss": info["loss"], "grad_norm": info["grad_norm"], "skipped": info["skipped"],
                      "config": {"workload": f"1 x {n} views x {Fr} frames x {lat}x{lat} latent, no CFG; motion_modules + i2v trainable"}}))


if __name__ == "__main__":
    main()
