"""Name-seeded synthetic weights shared by the golden generators and the tests that replay their vectors: a tensor is a
function of (case tag, parameter name, shape) only, so fixtures of real-width cases hold inputs and expected outputs but no
weights (TEST INFRASTRUCTURE; torch's CPU generator is deterministic for a given torch build, and the GPU box runs this image)."""
import ast
import hashlib
import os
import zlib

import torch

ORACLE_SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "unet_ref.py")


def oracle_fingerprint(path: str = ORACLE_SRC) -> str:
    """sha256 of the oracle's syntax tree with docstrings removed: changes with any edit of the oracle's CODE, not with comments or
    formatting.  The generators of the slow oracle goldens (gpu_tier_oracle.npz, config2_full.npz) store it next to the outputs and
    tests/test_oracle_golden.py asserts it — an oracle edit without regenerating them fails on the CPU tier, not as an unexplained
    parity miss on the GPU box."""
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef, ast.Module)) and node.body and \
                isinstance(node.body[0], ast.Expr) and isinstance(getattr(node.body[0], "value", None), ast.Constant) and \
                isinstance(node.body[0].value.value, str):
            node.body = node.body[1:] or [ast.Pass()]
    return hashlib.sha256(ast.dump(tree).encode()).hexdigest()


def weight_checksum(module: torch.nn.Module) -> float:
    """float64 sum over every parameter and buffer (in state_dict order): ties a stored oracle output to the weights it was computed with."""
    tot = 0.0
    for v in module.state_dict().values():
        tot += v.double().sum().item()
    return tot


def seeded_tensor(key: str, shape, scale: float = 1.0) -> torch.Tensor:
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    return torch.randn(*shape, generator=g) * scale


MIX2, MIX3 = [0.3], [0.3, -0.2, 0.5]


def fill_named(module: torch.nn.Module, tag: str, scale: float = 0.2):
    """Every parameter <- seeded_tensor(tag/name); mix_factor gets fixed non-trivial blends."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("mix_factor"):
                p.copy_(torch.tensor(MIX2 if p.numel() == 1 else MIX3))
            else:
                p.copy_(seeded_tensor(f"{tag}/{name}", p.shape, scale))
    return module
