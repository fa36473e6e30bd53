#!/usr/bin/env python
"""Oracle outputs for the two GPU-tier parity tests whose CPU side is too slow to recompute on every run (the GPU box spent
5.4 of its 12.5 minutes inside the first one): the fp32 CPU oracle UNet (oracle/unet_ref.py) is run ONCE here, its outputs are
committed as float16 (tests/golden/gpu_tier_oracle.npz, ~4 MB; the storage rounding 5e-4 sits below every bar that uses them)
and the tests re-draw weights and inputs from the same seeds through the functions of this file.

* ``sds5``  — BASELINE config 5: one 4D-SDS step (animate3d_amd.sds.sds_recon_loss, itself pinned by the reference's own function in
  tests/golden/sds.npz) over the oracle UNet at b = 1, 4 views x 16 frames, 32 x 32 latent, CFG batch of 8 videos, guidance 7.5:
  raw UNet output, reconstruction, latent gradient, loss.
* ``config1`` — BASELINE config 1: 1 view x 4 frames x 64 x 64 latent, no CFG: UNet output.

    python tests/golden/make_gpu_tier_goldens.py [--out FILE]         # ~6 min on 8 cores

The file also records a fingerprint of the oracle's code (tests/golden/seeded.py::oracle_fingerprint) and a float64 checksum of each
weight set; tests/test_oracle_golden.py asserts both on the CPU tier, so an oracle edit or a changed draw order cannot leave the
stored outputs silently stale: re-run this script (and make_config2_golden.py) in the same commit as any change of oracle/unet_ref.py.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import unet_ref as O  # noqa: E402
from tests.golden.seeded import oracle_fingerprint, weight_checksum  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_tier_oracle.npz")
SDS5 = dict(n=4, F=16, hw=(32, 32), b=1, seed_w=0, seed_in=2, t=500, guidance_scale=7.5, recon_std_rescale=0.5)
CONFIG1 = dict(n=1, F=4, hw=(64, 64), seed_w=3, seed_in=4)


def sds5_inputs():
    """Latents, noise, timestep, text / image embeddings and cameras of the config-5 SDS step (CPU tensors, fp32)."""
    from animate3d_amd.embeddings import get_camera
    n, F, hw, b = SDS5["n"], SDS5["F"], SDS5["hw"], SDS5["b"]
    g = torch.Generator().manual_seed(SDS5["seed_in"])
    lat = 0.18215 * 4 * torch.randn(b * n * F, 4, *hw, generator=g)
    noise = torch.randn(b, n, F - 1, 4, *hw, generator=g)
    t = torch.tensor([SDS5["t"]])
    text = torch.randn(2 * b * n, 77, 768, generator=g)
    emb = torch.randn(b * n, 1024, generator=g)
    c2w = get_camera(n).reshape(n, 1, 4, 4).expand(n, F, 4, 4).reshape(b * n * F, 4, 4).clone()
    return lat, noise, t, text, emb, c2w


def sds5_weights():
    return O.build_fast(O.UNetConfig(), SDS5["n"], SDS5["F"], SDS5["hw"], seed=SDS5["seed_w"])


def config1_weights():
    return O.build_fast(O.UNetConfig(), CONFIG1["n"], CONFIG1["F"], CONFIG1["hw"], seed=CONFIG1["seed_w"])


def config1_inputs():
    return O.synthetic_inputs(O.UNetConfig(), CONFIG1["n"], CONFIG1["n"], CONFIG1["F"], CONFIG1["hw"], seed=CONFIG1["seed_in"])


def main():
    from animate3d_amd.sds import sds_recon_loss
    t0 = time.time()
    out = {"oracle_sha256": np.array(oracle_fingerprint())}
    ref = sds5_weights()
    out["sds5_w_checksum"] = np.float64(weight_checksum(ref))
    lat, noise, t, text, emb, c2w = sds5_inputs()
    seen = {}

    def unet(*a, **k):
        y = ref(*a, **k)
        seen["sample"] = y.sample.detach().float()
        return y

    lr = lat.clone().requires_grad_(True)
    loss, aux = sds_recon_loss(unet, lr, t, text, emb, c2w, guidance_scale=SDS5["guidance_scale"], n_view=SDS5["n"], n_frame=SDS5["F"],
                               recon_std_rescale=SDS5["recon_std_rescale"], noise=noise)
    loss.backward()
    print(f"sds5 done at {time.time() - t0:.0f} s: loss {loss.item():.6e} |unet| max {seen['sample'].abs().max():.3f} "
          f"|grad| max {lr.grad.abs().max():.3e}", flush=True)
    # the gradient is tiny in absolute terms (mean-reduced loss): stored scaled so that float16 keeps its 11 bits
    gscale = float(2.0 ** -int(np.floor(np.log2(lr.grad.abs().max().item()))))
    out.update(sds5_unet=seen["sample"].numpy().astype(np.float16), sds5_recon=aux["latents_recon"].detach().float().numpy().astype(np.float16),
               sds5_grad_scaled=(lr.grad * gscale).numpy().astype(np.float16), sds5_grad_scale=np.float64(gscale),
               sds5_loss=np.float64(loss.item()), sds5_in_checksum=np.float64(lat.double().sum().item() + text.double().sum().item()))
    del ref
    ref = config1_weights()
    out["config1_w_checksum"] = np.float64(weight_checksum(ref))
    inp = config1_inputs()
    with torch.no_grad():
        y = ref(**inp).sample
    print(f"config1 done at {time.time() - t0:.0f} s: |y| max {y.abs().max():.3f}", flush=True)
    out.update(config1_sample=y.numpy().astype(np.float16), config1_in_checksum=np.float64(inp["sample"].double().sum().item()))
    dst = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else OUT
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
