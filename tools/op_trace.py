#!/usr/bin/env python
"""Per-op parity tracer: runs a UNet forward on the HIP op set and, for EVERY op call, recomputes the same op from the same inputs
with the plain-torch reference (tests/torch_ops.py, fp32 on the GPU), then prints the worst relative errors per op family.
Usage: python tools/op_trace.py [--fp16] [--smoke | --full]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from animate3d_amd.config import UNetConfig  # noqa: E402
from animate3d_amd.hip_ops import HipOps  # noqa: E402
from animate3d_amd.unet import MVUNetMotionModel  # noqa: E402
from oracle import unet_ref as O  # noqa: E402
from tests.torch_ops import TorchRefOps  # noqa: E402


class Tracer:
    def __init__(self, ops):
        self._ops, self._ref, self.rows = ops, TorchRefOps(torch.float32, "cuda"), []

    def __getattr__(self, name):
        fn = getattr(self._ops, name)
        if not callable(fn) or name.startswith("_") or name in ("empty", "interleave_geglu", "split_cols"):
            return fn
        rf = getattr(self._ref, name, None)

        def wrapped(*a, **k):
            k_ref = dict(k)
            if "out" in k and k["out"] is not None:
                k_ref["out"] = k["out"].float().clone()
            out = fn(*a, **k)
            if rf is not None:
                want = rf(*a, **k_ref)
                g, w = (out[0], want[0]) if isinstance(out, tuple) else (out, want)
                if torch.is_tensor(g) and g.dtype != torch.float64:
                    g, w = g.float(), w.float()
                    rel = ((g - w).norm() / (w.norm() + 1e-20)).item()
                    shp = tuple(a[0].shape) if torch.is_tensor(a[0]) else ()
                    extra = {kk: vv for kk, vv in k.items() if not torch.is_tensor(vv)}
                    self.rows.append((rel, name, shp, [x for x in a[1:] if not torch.is_tensor(x)][:8], extra))
            return out
        return wrapped


def main():
    fp16 = "--fp16" in sys.argv
    if "--full" in sys.argv:
        kw, build = {}, lambda cfg, n, F, hw: O.init_synthetic_weights(O.MVUNetMotionModelRef(cfg, n, F, hw).eval(), seed=0, dense=True)
    else:
        kw, build = dict(block_out_channels=(320, 640), down_has_attn=(True, False), layers_per_block=1), lambda cfg, n, F, hw: O.build_fast(cfg, n, F, hw, seed=0)
    n, F, hw, V = 2, 3, (16, 16), 2
    ocfg = O.UNetConfig(**kw)
    ref = build(ocfg, n, F, hw)
    dt = torch.float16 if fp16 else torch.bfloat16
    tr = Tracer(HipOps(act_dtype=dt))
    model = MVUNetMotionModel(UNetConfig(**kw), num_views=n, device="cuda", ops=tr)
    model.load_state_dict(ref.state_dict(), strict=True)
    model = model.to(dt).eval()
    inp = O.synthetic_inputs(ocfg, V, n, F, hw, seed=1)
    ci = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    ci["added_cond_kwargs"] = {"image_embeds": inp["added_cond_kwargs"]["image_embeds"].cuda()}
    y = model(**ci).sample
    y_ref = ref(**inp).sample
    print(f"end to end ({dt}): rel L2 vs oracle {((y.float().cpu() - y_ref).norm() / y_ref.norm()).item():.3e}")
    worst = {}
    for r in tr.rows:
        if r[1] not in worst or r[0] > worst[r[1]][0]:
            worst[r[1]] = r
    for name, r in sorted(worst.items(), key=lambda kv: -kv[1][0]):
        print(f"{name:16s} worst rel {r[0]:.3e}  x{r[2]} args {r[3]} {r[4]}")
    print("-- top 12 calls")
    for r in sorted(tr.rows, key=lambda r: -r[0])[:12]:
        print(f"{r[1]:16s} rel {r[0]:.3e}  x{r[2]} args {r[3]} {r[4]}")


if __name__ == "__main__":
    main()
