"""Which host thread count gives the CPU oracle its best BASELINE-config-1 time on the GPU box (calibrates bench.py's cpu_baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unet_ref as O
cfg = O.UNetConfig()
ref = O.build_fast(cfg, 1, 4, (64, 64), seed=0)
inp = O.synthetic_inputs(cfg, 1, 1, 4, (64, 64), seed=1)
print("host logical CPUs:", os.cpu_count())
for th in (16, 32, 64, 96, 128, 192, 256):
    if th > (os.cpu_count() or 1):
        continue
    torch.set_num_threads(th)
    ref(**inp)
    t0 = time.time(); ref(**inp); dt = time.time() - t0
    print(f"threads {th:4d}: {dt:7.2f} s/forward", flush=True)
