#!/usr/bin/env python
"""Race / miscount detector for the LDS-DMA staged attention backward (csrc/attn_bwd.hip, round 6): seeded cases at every head dim and both
key maps, repeated under unrelated HBM traffic; prints one sha256 per (case, repeat) of dQ | dK | dV.  Run once with the in-tree library and
once with a side build that keeps the register staging (python -m animate3d_amd.build --experiment A3D_EXP_BWD_NODMA; A3D_LIB=...): the two
stagings move the same bytes into the same LDS image, so every line must be identical."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import animate3d_amd.hip_ops as _hip_ops  # noqa: E402
from animate3d_amd.hip_ops import HipOps, RowMap  # noqa: E402

if os.environ.get("A3D_LIB"):
    _hip_ops._LIB_PATH = os.environ["A3D_LIB"]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    junk = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    for dt in (torch.bfloat16, torch.float16):
        ops = HipOps(act_dtype=dt)
        for (D, n, F, L, b) in ((40, 4, 4, 1024, 1), (40, 2, 3, 256, 2), (80, 4, 4, 256, 1), (80, 2, 2, 128, 2), (160, 4, 4, 64, 1), (160, 2, 2, 32, 3)):
            heads, C = 8, 8 * D
            rows = b * n * F * L
            g = torch.Generator(device="cuda").manual_seed(D * 1000 + L)
            kvq = (torch.randn(rows, 3 * C, generator=g, device="cuda")).to(dt)
            q, k, v = kvq[:, 2 * C:], kvq[:, :C], kvq[:, C:2 * C]
            do = torch.randn(rows, C, generator=g, device="cuda").to(dt)
            qm = RowMap(F, n * F * L, L, L, F * L)
            k0 = RowMap(F, n * F * L, 0, L, F * L)
            S, G = n * L, b * F
            for name, km, share in (("mv", qm, 1), ("ff", k0, F)):
                o, lse = ops.flash_attn(q, k, v, qm, km, G, heads, S, S, with_lse=True)
                for r in range(reps):
                    if r % 2:
                        junk.add_(1)
                    dq, dk, dv = ops.flash_attn_bwd(q, k, v, do, qm, km, G, heads, S, S, q_per_kv=share, o=o, lse=lse)
                    torch.cuda.synchronize()
                    h = hashlib.sha256()
                    for t in (dq, dk, dv):
                        h.update(t.contiguous().view(torch.int16).cpu().numpy().tobytes())
                    print(f"{str(dt)[6:]} D={D} S={S} G={G} {name} rep{r} {h.hexdigest()[:16]}", flush=True)


if __name__ == "__main__":
    main()
