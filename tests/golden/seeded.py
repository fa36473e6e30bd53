"""Name-seeded synthetic weights shared by the golden generators and the tests that replay their vectors: a tensor is a
function of (case tag, parameter name, shape) only, so fixtures of real-width cases hold inputs and expected outputs but no
weights (TEST INFRASTRUCTURE; torch's CPU generator is deterministic for a given torch build, and the GPU box runs this image)."""
import zlib

import torch


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
