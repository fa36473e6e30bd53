"""Workload for PMC passes over the implicit-GEMM 3x3 convolution of the persistent kernel next to the DENSE persistent GEMM at the same
M x N x K (round 5): level 0 (B = 128, 64 x 64, 320 -> 320, + residual: M = 524288, N = 320, K = 2880), level 1
(32 x 32, 640 -> 640, + residual: M = 131072, N = 640, K = 5760), level 3 (8 x 8, 1280 -> 1280: 128 tiles on 256 CUs); three launches each,
conv and dense interleaved so that both see the same clock state.  A3D_LIB selects the library build (A/B of two builds)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import animate3d_amd.hip_ops as _hip_ops  # noqa: E402
from animate3d_amd.hip_ops import HipOps  # noqa: E402

if os.environ.get("A3D_LIB"):
    _hip_ops._LIB_PATH = os.environ["A3D_LIB"]

ops = HipOps()
bf = torch.bfloat16
rnd = lambda *s, scale=1.0: (torch.randn(*s, device="cuda") * scale).to(bf)
cases = []
for (B, H, Cin, Cout) in ((128, 64, 320, 320), (128, 32, 640, 640), (128, 8, 1280, 1280)):
    M, K = B * H * H, 9 * Cin
    cases.append((B, H, Cin, Cout, rnd(M, Cin), rnd(Cout, K, scale=K ** -0.5), torch.zeros(Cout, device="cuda"), rnd(M, Cout), rnd(M, K)))
for _ in range(3):
    for (B, H, Cin, Cout, x, w, b, r, xd) in cases:
        ops.conv3x3(x, B, H, H, w, b, residual=r)
        ops.gemm(xd, w, b, residual=r)
torch.cuda.synchronize()
print("done")
