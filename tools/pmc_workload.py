"""Workload for the rocprofv3 PMC passes behind bench.py's roofline.traffic (profiles/r2_flash_pmc_traffic.json): one calibration
stream of known size (torch x * 1.5 on 2 GB of fp32: reads 2.013 GB, writes 2.013 GB) and a few launches of the dominant kernel at the
level-0 shape of BASELINE config 2 (32 groups x 8 heads x 16384 queries x 16384 keys, head_dim 40; Q / K / V are column slices of
one [524288, 960] bf16 buffer as in the UNet).  `--exact` / `--plain` profile the LDS-DMA kernel's exact pass / the generic kernel instead."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from animate3d_amd.hip_ops import HipOps, RowMap  # noqa: E402

kw = dict(exact="--exact" in sys.argv, plain="--plain" in sys.argv)
ops = HipOps()
x = torch.ones(503316480, device="cuda", dtype=torch.float32)
y = x * 1.5                                   # calibration kernel
torch.cuda.synchronize()
del x, y
D, n, F, L, b, heads = 40, 4, 16, 4096, 2, 8
C = heads * D
qkv = (torch.randn(b * n * F * L, 3 * C, device="cuda") * 1.0).to(torch.bfloat16)
qm = RowMap(F, n * F * L, L, L, F * L)
for _ in range(4):
    o = ops.flash_attn(qkv[:, 2 * C:], qkv[:, :C], qkv[:, C:2 * C], qm, qm, b * F, heads, n * L, n * L, **kw)
torch.cuda.synchronize()
print("done", float(o.float().abs().mean()))
