"""Generate tests/golden/from_unet2d.json by running the REFERENCE's own ``MVUNetMotionModel.from_unet2d`` /
``load_motion_modules``.

Run in the build container only (needs /root/reference):

    python -B tests/golden/make_from_unet2d_goldens.py

The two methods (animatediff/models/unet_motion_mv_model.py:275-368, 394-402) are compiled from the reference file's syntax tree
as it lies and run on three differently seeded instances of the oracle's module tree (oracle/unet_ref.py, diffusers' parameter
names): one plays the 2-D MVDream UNet, one the MotionAdapter, one is what ``cls.from_config`` returns.  Afterwards every
parameter of the result is classified by where its value came from.  The fixture is that classification (parameter name ->
"unet" | "adapter" | "untouched" | "same"): the weight mapping a drop-in ``from_unet2d`` has to reproduce.

Only data (parameter names of the diffusers layout and their provenance) is written; no reference source leaves /root/reference.
"""
import ast
import json
import os
import sys
import types
from typing import Optional

sys.dont_write_bytecode = True
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import unet_ref as O  # noqa: E402

REF = "/root/reference"
SMALL = dict(block_out_channels=(32, 64, 64, 64), num_attention_heads=4, norm_num_groups=8)


def build(seed):
    m = O.MVUNetMotionModelRef(O.UNetConfig(**SMALL), 2, 3, (8, 8)).eval()
    O.init_synthetic_weights(m, seed=seed, dense=True)
    m.time_proj, m.conv_act, m.dtype = nn.Identity(), nn.SiLU(), torch.float32      # parameter-free members the method touches
    return m


def main():
    tree = ast.parse(open(os.path.join(REF, "animatediff/models/unet_motion_mv_model.py")).read())
    cls = next(nd for nd in tree.body if isinstance(nd, ast.ClassDef) and nd.name == "MVUNetMotionModel")
    fns = [nd for nd in cls.body if isinstance(nd, ast.FunctionDef) and nd.name in ("from_unet2d", "load_motion_modules")]
    for nd in fns:
        nd.decorator_list = []
    ns = {"torch": torch, "Optional": Optional, "MVUNet2DConditionModel": object, "MotionAdapter": object}
    exec(compile(ast.fix_missing_locations(ast.Module(body=fns, type_ignores=[])), "from_unet2d", "exec"), ns)

    unet, adapter, fresh = build(1), build(2), build(3)
    before = {k: v.clone() for k, v in fresh.state_dict().items()}
    unet.config = {"down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"], "up_block_types": ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3,
                   "num_attention_heads": None, "attention_head_dim": 4}
    adapter.config = {"motion_num_attention_heads": 4, "motion_max_seq_length": 32, "use_motion_mid_block": True, "conv_in_channels": None}
    fresh.load_motion_modules = types.MethodType(ns["load_motion_modules"], fresh)
    fake_cls = types.SimpleNamespace(__name__="MVUNetMotionModel", from_config=lambda config: fresh)
    model = ns["from_unet2d"](fake_cls, unet, adapter)
    assert model is fresh
    assert unet.config["down_block_types"] == ["CrossAttnDownBlockMotion"] * 3 + ["DownBlockMotion"] and unet.config["num_attention_heads"] == 4
    a, b, tags = unet.state_dict(), adapter.state_dict(), {}
    for k, v in fresh.state_dict().items():
        if torch.equal(v, a[k]) and torch.equal(v, before[k]):
            tags[k] = "same"                      # seed-independent buffers (sinusoidal tables)
        elif torch.equal(v, a[k]):
            tags[k] = "unet"
        elif torch.equal(v, b[k]):
            tags[k] = "adapter"
        else:
            assert torch.equal(v, before[k]), k
            tags[k] = "untouched"
    path = os.path.join(HERE, "from_unet2d.json")
    with open(path, "w") as f:
        json.dump(tags, f, indent=0, sort_keys=True)
    from collections import Counter
    print("wrote", path, Counter(tags.values()))
    print("untouched prefixes:", sorted({k.split(".")[0] for k, t in tags.items() if t == "untouched"}))


if __name__ == "__main__":
    main()
