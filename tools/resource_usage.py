#!/usr/bin/env python
"""Per-kernel register / spill / LDS summary of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), gfx950.
Usage: python tools/resource_usage.py animate3d_amd/csrc/gemm_conv.hip [name-filter] [-DFLAG ...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
defs = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *defs, "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r": remark: (.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    txt = m.group(1).strip()
    if txt.startswith("Function Name:") or txt.startswith("Name:"):
        cur = txt.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in txt:
        k, v = txt.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("(anonymous namespace)::", "").replace("void ", "")
    if flt and flt not in dem:
        continue
    print(f"{dem[:70]:70s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} spill {r.get('VGPRs Spill','?'):>3s} "
          f"sgpr-spill {r.get('SGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?')}")
