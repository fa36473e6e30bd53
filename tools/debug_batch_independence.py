"""Find the first op whose output for the first CFG half depends on the presence of the second half (16x16 latent, 4 videos = 2 x 2 views,
2 frames): every op output of the full-batch run is compared with the half-batch run's, row-wise (b is the outermost row index everywhere)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from animate3d_amd.config import UNetConfig  # noqa: E402
from animate3d_amd.hip_ops import HipOps  # noqa: E402
from animate3d_amd.unet import MVUNetMotionModel  # noqa: E402
from oracle import unet_ref as O  # noqa: E402

NAMES = ("gemm", "gemm_geglu", "conv3x3", "flash_attn", "flash_attn2", "temporal_attn", "group_norm", "layer_norm", "concat", "timestep_embed",
         "im2col_in", "unpack_out", "gemm_f32out")


class Rec:
    def __init__(self, ops):
        self._ops, self.log = ops, []

    def __getattr__(self, name):
        fn = getattr(self._ops, name)
        if name not in NAMES:
            return fn

        def wrapped(*a, **k):
            out = fn(*a, **k)
            first = out[0] if isinstance(out, tuple) else out
            desc = name + " " + " ".join(str(tuple(t.shape)) for t in a if torch.is_tensor(t))[:90] + (" out=view" if k.get("out") is not None else "")
            self.log.append((desc, None if first is None else first.detach().float().clone()))
            if isinstance(out, tuple) and len(out) == 2 and torch.is_tensor(out[1]):
                self.log.append((desc + " [2nd]", out[1].detach().float().clone()))
            return out
        return wrapped

    def __setattr__(self, name, value):
        if name in ("_ops", "log"):
            object.__setattr__(self, name, value)
        else:
            setattr(self._ops, name, value)


n, F, videos, hw = 2, 2, 4, (16, 16)
cfg = UNetConfig()
rec = Rec(HipOps())
model = MVUNetMotionModel(cfg, ops=rec, num_views=n, device="cuda")
model.init_synthetic(seed=0)
model = model.to(torch.bfloat16).eval()
rec._ops.split_k = ("--split" in sys.argv)
inp = O.synthetic_inputs(O.UNetConfig(), videos, n, F, hw, seed=11, cfg_doubled=True)
inp = {k: (v.cuda() if torch.is_tensor(v) else ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v)) for k, v in inp.items()}
model(**inp)                      # weight packing (its folding GEMMs) outside the recording
rec.log = []
full = model(**inp).sample
log_full, rec.log = rec.log, []
half = dict(inp)
for k in ("sample", "encoder_hidden_states", "camera"):
    half[k] = inp[k][: videos // 2]
half["added_cond_kwargs"] = {"image_embeds": inp["added_cond_kwargs"]["image_embeds"][: videos // 2]}
part = model(**half).sample
log_half = rec.log
print("final:", "equal" if torch.equal(full[: videos // 2], part) else f"DIFFERENT max {(full[: videos // 2] - part).abs().max().item():.3e}", len(log_full), len(log_half))
bad = 0
for i, ((d1, a), (d2, b)) in enumerate(zip(log_full, log_half)):
    if a is None or b is None:
        continue
    if a.dim() == 5:
        a1 = a[: b.shape[0]]
    else:
        a1 = a[: b.shape[0]]
    if a1.shape != b.shape:
        print(i, "shape mismatch", d1, "|", d2, tuple(a.shape), tuple(b.shape))
        continue
    if not torch.equal(a1, b):
        print(f"op {i}: {d1}  ||  {d2}: max diff {(a1 - b).abs().max().item():.3e} of {b.abs().max().item():.3e}")
        bad += 1
        if bad >= 6:
            break
