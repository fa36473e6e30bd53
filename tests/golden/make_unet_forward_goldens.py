"""Generate tests/golden/unet_forward.npz by running the REFERENCE's own ``MVUNetMotionModel.forward``.

Run in the build container only (needs /root/reference):

    python -B tests/golden/make_unet_forward_goldens.py

``MVUNetMotionModel.forward`` (animatediff/models/unet_motion_mv_model.py:633-867) is compiled from the reference file's syntax
tree as it lies (the module itself imports diffusers, absent here) and called on an object that exposes, under the attribute
names and call signatures that method uses, the restated third-party pieces of oracle/unet_ref.py: ``time_proj`` /
``time_embedding`` / ``camera_embedding`` / ``encoder_hid_proj`` / ``conv_in`` / the down, mid and up blocks (diffusers'
CrossAttnDownBlockMotion & co. as restated, with the reference's processors installed) / ``conv_norm_out`` / ``conv_act`` /
``conv_out``.  What the vectors pin is the reference-owned top-level glue of the denoise step: the timestep / camera embedding
sum, the frame-0 time-zero embedding of ``i2v_cond_time_zero``, the per-frame repeat of text and IP tokens, the
``(V, C, F, h, w) <-> ((V F), C, h, w)`` folds, the skip-connection bookkeeping and the forced-upsample-size rule.

Only data (seeded inputs and the reference's outputs) is written; no reference source leaves /root/reference.
"""
import ast
import os
import sys
from types import SimpleNamespace
from typing import Any, Dict, Optional, Tuple, Union

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn.functional as F
from einops import rearrange

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import unet_ref as O  # noqa: E402

REF = "/root/reference"
SMALL = dict(block_out_channels=(32, 64, 64, 64), num_attention_heads=4, norm_num_groups=8)
CASES = [("scalar_t", dict(n=2, videos=4, F=3, hw=(8, 8), tz=False, tensor_t=False)),
         ("time_zero", dict(n=2, videos=4, F=3, hw=(8, 8), tz=True, tensor_t=True)),
         ("odd_size", dict(n=2, videos=2, F=2, hw=(12, 20), tz=True, tensor_t=False))]      # 12 and 20 are not multiples of 8: forced upsample sizes


def reference_forward():
    tree = ast.parse(open(os.path.join(REF, "animatediff/models/unet_motion_mv_model.py")).read())
    cls = next(nd for nd in tree.body if isinstance(nd, ast.ClassDef) and nd.name == "MVUNetMotionModel")
    fwd = next(nd for nd in cls.body if isinstance(nd, ast.FunctionDef) and nd.name == "forward")
    fwd.decorator_list = []
    ns = {"torch": torch, "rearrange": rearrange, "Union": Union, "Optional": Optional, "Dict": Dict, "Any": Any, "Tuple": Tuple,
          "logger": SimpleNamespace(info=lambda *a, **k: None),
          "UNet3DConditionOutput": lambda sample: SimpleNamespace(sample=sample)}
    exec(compile(ast.fix_missing_locations(ast.Module(body=[fwd], type_ignores=[])), "unet_forward", "exec"), ns)
    return ns["forward"]


class _Down:
    def __init__(self, blk):
        self.blk, self.has_cross_attention = blk, blk.has_cross_attention

    def __call__(self, hidden_states, temb, encoder_hidden_states=None, attention_mask=None, num_frames=1, cross_attention_kwargs=None):
        return self.blk(hidden_states, temb, encoder_hidden_states, num_frames)


class _Mid:
    def __init__(self, blk):
        self.blk, self.motion_modules = blk, blk.motion_modules

    def __call__(self, hidden_states, temb, encoder_hidden_states=None, attention_mask=None, num_frames=1, cross_attention_kwargs=None):
        return self.blk(hidden_states, temb, encoder_hidden_states, num_frames)


class _Up:
    def __init__(self, blk):
        self.blk, self.has_cross_attention, self.resnets = blk, blk.has_cross_attention, blk.resnets

    def __call__(self, hidden_states, temb, res_hidden_states_tuple, encoder_hidden_states=None, upsample_size=None, attention_mask=None,
                 num_frames=1, cross_attention_kwargs=None):
        return self.blk(hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states, num_frames, upsample_size)


def as_reference_self(m):
    dim0 = m.cfg.block_out_channels[0]
    return SimpleNamespace(num_upsamplers=m.num_upsamplers, dtype=torch.float32, config=m.config, __class__=type(m),
                           time_proj=lambda t: O.timestep_sinusoid(t, dim0), time_embedding=lambda e, cond=None: m.time_embedding(e),
                           camera_embedding=m.camera_embedding, encoder_hid_proj=m.encoder_hid_proj, conv_in=m.conv_in,
                           down_blocks=[_Down(b) for b in m.down_blocks], mid_block=_Mid(m.mid_block), up_blocks=[_Up(b) for b in m.up_blocks],
                           conv_norm_out=m.conv_norm_out, conv_act=F.silu, conv_out=m.conv_out)


def main():
    fwd = reference_forward()
    out = {}
    for tag, c in CASES:
        cfg = O.UNetConfig(**SMALL)
        m = O.MVUNetMotionModelRef(cfg, c["n"], c["F"], c["hw"]).eval()
        O.init_synthetic_weights(m, seed=0, dense=True)
        inp = O.synthetic_inputs(cfg, c["videos"], c["n"], c["F"], c["hw"], seed=13, cfg_doubled=c["videos"] >= 2 * c["n"])
        if c["tensor_t"]:
            inp["timestep"] = torch.tensor([501, 37, 999, 4][: c["videos"]])
        with torch.no_grad():
            y = fwd(as_reference_self(m), inp["sample"], inp["timestep"], inp["encoder_hidden_states"], added_cond_kwargs=inp["added_cond_kwargs"],
                    camera=inp["camera"], num_views=c["n"], i2v_cond_time_zero=c["tz"]).sample
            mine = m(**inp, i2v_cond_time_zero=c["tz"]).sample
        print(tag, tuple(y.shape), "oracle forward vs reference forward: max |diff|", (y - mine).abs().max().item(), "of", y.abs().max().item())
        out[f"{tag}/y"] = y.numpy().copy()                       # inputs and weights are re-derived from their seeds by the test
        out[f"{tag}/sample"] = inp["sample"].numpy().copy()      # (kept as a guard against a change of the seeded generators)
        out[f"{tag}/timestep"] = np.asarray(inp["timestep"])
        out[f"{tag}/cfg"] = np.array([c["n"], c["videos"], c["F"], c["hw"][0], c["hw"][1], float(c["tz"])])
    path = os.path.join(HERE, "unet_forward.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
