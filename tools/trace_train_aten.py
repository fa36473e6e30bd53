"""Which eager-torch kernels does one training step still launch, and from where?

Runs ``animate3d_amd.train.training_step`` (train.yaml shape, as tools/bench_train.py) under a ``TorchDispatchMode`` and aggregates every aten
op that touches a device tensor by (op, innermost animate3d_amd source line | "autograd engine") with call counts and the bytes of its
outputs.  The C-ABI kernels are invisible here (ctypes launches): what is listed is exactly the ``at::native`` tail of
``profiles/r*_train_kernel_stats.md``.  Usage: python tools/trace_train_aten.py [--top 50]"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SKIP = ("aten.view", "aten._unsafe_view", "aten.slice", "aten.select", "aten.expand", "aten.reshape", "aten.t.", "aten.transpose", "aten.permute",
        "aten.detach", "aten.alias", "aten.as_strided", "aten.unsqueeze", "aten.squeeze", "aten.empty", "aten.unbind", "aten.split", "aten.lift_fresh",
        "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten._local_scalar_dense", "aten.new_empty", "aten.narrow", "aten.unfold", "aten.chunk")


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.defaultdict(lambda: [0, 0])

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith(SKIP):
            return out
        outs = out if isinstance(out, (tuple, list)) else (out,)
        nbytes = sum(o.numel() * o.element_size() for o in outs if torch.is_tensor(o) and o.is_cuda)
        if nbytes == 0 and not any(torch.is_tensor(a) and a.is_cuda for a in args):
            return out
        where = "autograd engine / torch internals"
        for fr in reversed(traceback.extract_stack(limit=40)[:-1]):
            if "animate3d_amd" in fr.filename:
                where = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                break
        r = self.rows[(name, where)]
        r[0] += 1
        r[1] += nbytes
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    from animate3d_amd.config import UNetConfig
    from animate3d_amd.denoise import ddim_schedule
    from animate3d_amd.embeddings import get_camera
    from animate3d_amd.train import FlatAdamW, select_trainable, training_step
    from animate3d_amd.unet import MVUNetMotionModel
    n, Fr, lat = 4, 16, 32
    model = MVUNetMotionModel(UNetConfig(), num_views=n, device="cuda").init_synthetic(seed=0)
    params = select_trainable(model)
    model.enable_training(compute_dtype=torch.bfloat16)
    opt = FlatAdamW(params, model.ops, lr=1e-4, loss_scale=None)
    g = torch.Generator(device="cuda").manual_seed(0)
    latents = torch.randn(1, n, 4, Fr, lat, lat, generator=g, device="cuda") * 0.5
    text = torch.randn(1, 77, 768, generator=g, device="cuda")
    cams = get_camera(n).cuda()
    img = torch.randn(n, 1024, generator=g, device="cuda")
    ac = ddim_schedule(25)[1]
    step = lambda: training_step(model, opt, latents, text, cams, img, alphas_cumprod=ac, num_views=n, generator=g)
    step()
    torch.cuda.synchronize()
    tr = Trace()
    with tr:
        step()
    torch.cuda.synchronize()
    rows = sorted(tr.rows.items(), key=lambda kv: -kv[1][1])
    tot_calls = sum(v[0] for v in tr.rows.values())
    tot_bytes = sum(v[1] for v in tr.rows.values())
    print(f"# one training step: {tot_calls} eager aten ops on device tensors, {tot_bytes / 2 ** 30:.2f} GiB of outputs")
    print("| output MiB | calls | op | from |")
    print("|---|---|---|---|")
    for (name, where), (calls, nbytes) in rows[:args.top]:
        print(f"| {nbytes / 2 ** 20:9.1f} | {calls:5d} | `{name}` | {where} |")
    by_op = collections.defaultdict(lambda: [0, 0])
    for (name, _), (calls, nbytes) in tr.rows.items():
        by_op[name][0] += calls
        by_op[name][1] += nbytes
    print("\n| output MiB | calls | op (all call sites) |")
    print("|---|---|---|")
    for name, (calls, nbytes) in sorted(by_op.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"| {nbytes / 2 ** 20:9.1f} | {calls:5d} | `{name}` |")


if __name__ == "__main__":
    main()
