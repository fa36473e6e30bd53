"""Workload for PMC passes over the persistent GEMM at the small-K level-0 shapes (M = 524288): a few launches each of
N = 1280 / K = 320 (linear), N = 960 / K = 320, N = 320 / K = 320 + residual, fused GEGLU N = 2560 / K = 320, N = 1920 / K = 640 (M = 131072)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from animate3d_amd.hip_ops import HipOps  # noqa: E402

ops = HipOps()
bf = torch.bfloat16
M = 524288
x = (torch.randn(M, 320, device="cuda")).to(bf)
r = (torch.randn(M, 320, device="cuda")).to(bf)
w1280 = (torch.randn(1280, 320, device="cuda") * 0.05).to(bf)
w960 = (torch.randn(960, 320, device="cuda") * 0.05).to(bf)
w320 = (torch.randn(320, 320, device="cuda") * 0.05).to(bf)
wg = (torch.randn(2560, 320, device="cuda") * 0.05).to(bf)
bg = torch.zeros(2560, device="cuda")
x2 = (torch.randn(131072, 640, device="cuda")).to(bf)
w1920 = (torch.randn(1920, 640, device="cuda") * 0.04).to(bf)
for _ in range(3):
    ops.gemm(x, w1280)
    ops.gemm(x, w960)
    ops.gemm(x, w320, residual=r)
    ops.gemm_geglu(x, wg, bg)
    ops.gemm(x2, w1920)
torch.cuda.synchronize()
print("done")
