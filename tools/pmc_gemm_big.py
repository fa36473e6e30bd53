"""Workload for TCC / TCP PMC passes over the persistent GEMM at large-K shapes (full chip): 3 launches each of
M = 32768, N = 1280, K = 5120 (NB = 5), 8192^3 (NB = 4) and the 3x3 conv 16x16 2560 -> 1280."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from animate3d_amd.hip_ops import HipOps  # noqa: E402

ops = HipOps()
bf = torch.bfloat16
x = torch.randn(32768, 5120, device="cuda").to(bf)
w = (torch.randn(1280, 5120, device="cuda") * 0.02).to(bf)
x8 = (torch.rand(8192, 8192, device="cuda") * 2 - 1).to(bf)
w8 = (torch.rand(8192, 8192, device="cuda") * 2 - 1).to(bf)
for _ in range(3):
    ops.gemm(x, w)
    ops.gemm(x8, w8)
torch.cuda.synchronize()
print("done")
