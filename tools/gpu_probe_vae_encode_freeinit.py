"""GPU probe of the two pieces added after round 1's GPU budget was spent (VAE encoder, FreeInit mix): NOT YET RUN on an MI355X.
Run it first in the next round (`python tools/gpu_probe_vae_encode_freeinit.py`), then turn it into `-m gpu` tests in
tests/test_vae.py / tests/test_denoise_loop.py with the tolerances it reports.  Expected: relative L2 of the posterior moments
≈ 1.4e-2 (mean) / 2e-2 (logvar) — what the bf16-storage emulation of the same host logic (tests/torch_ops.py with bfloat16) gives
on CPU; bar 3e-2 (the decoder's), FreeInit mix equal to the CPU result to fp32 FFT rounding."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animate3d_amd.vae import AutoencoderKLEncoder
from animate3d_amd.denoise import free_init_filter, free_init_renoise
from oracle import vae_ref as R

t0 = time.time()
ref = R.init_synthetic_weights(R.VAEEncoderRef(), seed=1).eval()
enc = AutoencoderKLEncoder(device="cuda")
enc.load_state_dict(ref.state_dict(), strict=True)
enc = enc.to(torch.bfloat16).eval()
for hw in ((64, 64), (64, 96)):          # latent 8x8 (64 tokens) and 8x12 (96 tokens: the zero-padded P V contraction)
    x = torch.rand(2, 3, *hw, generator=torch.Generator().manual_seed(3)) * 2 - 1
    mw, lw = ref.encode(x)
    mg, lg = enc.encode(x.cuda())
    torch.cuda.synchronize()
    rel_m = ((mg.cpu() - mw).norm() / mw.norm()).item()
    rel_l = ((lg.cpu() - lw).norm() / lw.norm()).item()
    print(f"[probe] VAE encode 2x3x{hw[0]}x{hw[1]} real widths: rel_l2 mean {rel_m:.3e} logvar {rel_l:.3e} shape {tuple(mg.shape)} "
          f"finite {bool(torch.isfinite(mg).all())}")
g = torch.Generator().manual_seed(4)
rest, init = torch.randn(2, 4, 15, 16, 16, generator=g), torch.randn(2, 4, 15, 16, 16, generator=g)
filt = free_init_filter((1, 4, 15, 16, 16))
want = free_init_renoise(rest, init, 0.0047, filt, torch.Generator().manual_seed(9))
gg = torch.Generator(device="cuda").manual_seed(9)
got = free_init_renoise(rest.cuda(), init.cuda(), 0.0047, filt.cuda(), gg)
# different RNG streams on CPU and GPU: compare the deterministic part (same noise injected)
from animate3d_amd.denoise import free_init_mix
z = torch.randn(2, 4, 15, 16, 16, generator=g)
d = (free_init_mix(rest.cuda(), z.cuda(), filt.cuda()).cpu() - free_init_mix(rest, z, filt)).abs().max().item()
print(f"[probe] FreeInit mix cuda vs cpu max|diff| {d:.3e}; renoise finite {bool(torch.isfinite(got).all())} std {got.std().item():.3f} (cpu {want.std().item():.3f})")
print(f"[probe] {time.time() - t0:.1f} s")
