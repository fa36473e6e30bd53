"""VAE decode and encode: host logic on CPU with the plain-torch op set, HIP kernels on the GPU (decode),
both against the oracle restatement of diffusers' AutoencoderKL (oracle/vae_ref.py; parity unpinned — no diffusers offline)."""
import numpy as np
import pytest
import torch

from animate3d_amd.vae import AutoencoderKLDecoder, AutoencoderKLEncoder, VAEConfig
from oracle import vae_ref as R
from tests.torch_ops import TorchRefOps

SMALL = dict(block_out_channels=(32, 64, 64, 64), attention_head_dim=64)


def test_state_dict_keys_and_host_logic_match_oracle():
    ref = R.init_synthetic_weights(R.VAEDecoderRef(R.VAEConfig(**SMALL)), seed=0).eval()
    vae = AutoencoderKLDecoder(VAEConfig(**SMALL), ops=TorchRefOps())
    assert list(vae.state_dict().keys()) == list(ref.state_dict().keys())
    missing, unexpected = vae.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected
    lat = torch.randn(2, 4, 3, 6, 4, generator=torch.Generator().manual_seed(1)) * 0.18215
    want = ref.decode_latents(lat)
    got = vae.decode_latents(lat)
    assert got.shape == want.shape == (2, 3, 3, 48, 32) and got.dtype == torch.float32
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-3, atol=2e-4)
    z = torch.randn(3, 4, 4, 4)
    np.testing.assert_allclose(vae.decode(z).numpy(), ref.decode(z).numpy(), rtol=2e-3, atol=2e-4)


def test_sd15_vae_decoder_key_count():
    """The SD1.5 AutoencoderKL decoder half has 138 tensors under `decoder.*` + 2 under `post_quant_conv.*` (structural KAT:
    15 resnets x (8 + 2 shortcut for the two channel changes) + attention 10 + 3 upsamplers x 2 + conv_in/out/norm_out 6)."""
    vae = AutoencoderKLDecoder(device="meta")
    keys = list(vae.state_dict().keys())
    assert len([k for k in keys if k.startswith("decoder.")]) == 138 and len(keys) == 140
    assert vae.decoder.up_blocks[2].resnets[0].conv_shortcut.weight.shape == (256, 512, 1, 1)
    assert vae.decoder.up_blocks[3].upsamplers is None and vae.decoder.conv_out.weight.shape == (3, 128, 3, 3)
    with pytest.raises(ValueError):
        AutoencoderKLDecoder(VAEConfig(attention_head_dim=64), device="meta")      # multi-head mid attention: not the SD VAE


def test_encoder_state_dict_keys_and_host_logic_match_oracle():
    """Encoder half on the plain-torch op set: same keys as the oracle's (= diffusers') modules, the flipped stride-2 conv
    reproduces Downsample2D's right/bottom padding, quant_conv + clamp + sampling follow pipeline.py:556-560."""
    ref = R.init_synthetic_weights(R.VAEEncoderRef(R.VAEConfig(**SMALL)), seed=1).eval()
    enc = AutoencoderKLEncoder(VAEConfig(**SMALL), ops=TorchRefOps())
    assert list(enc.state_dict().keys()) == list(ref.state_dict().keys())
    missing, unexpected = enc.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected
    x = torch.rand(2, 3, 32, 48, generator=torch.Generator().manual_seed(3)) * 2 - 1
    mean_w, logvar_w = ref.encode(x)
    mean_g, logvar_g = enc.encode(x)
    assert mean_g.shape == mean_w.shape == (2, 4, 4, 6) and mean_g.dtype == torch.float32
    np.testing.assert_allclose(mean_g.numpy(), mean_w.numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(logvar_g.numpy(), logvar_w.numpy(), rtol=2e-3, atol=2e-4)
    want = ref.encode_latents(x, generator=torch.Generator().manual_seed(7))
    got = enc.encode_latents(x, generator=torch.Generator().manual_seed(7))
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-3, atol=2e-4)
    with pytest.raises(ValueError):
        enc.encode(torch.zeros(1, 3, 36, 32))                                        # not a multiple of 8
    with pytest.raises(ValueError):
        enc.encode(torch.zeros(1, 4, 32, 32))


def test_flipped_stride2_conv_is_the_right_bottom_padded_conv():
    """The identity the encoder's downsampler rests on, on odd-looking sizes and a non-symmetric filter:
    conv(F.pad(x, (0,1,0,1)), w, stride 2) == flip(conv(flip(x), flip(w), stride 2, padding 1)) for even H, W."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    x, w, b = torch.randn(2, 5, 6, 10, generator=g), torch.randn(7, 5, 3, 3, generator=g), torch.randn(7, generator=g)
    want = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    got = F.conv2d(x.flip(2, 3), w.flip(2, 3), b, stride=2, padding=1).flip(2, 3)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_sd15_vae_encoder_key_count():
    """SD1.5 AutoencoderKL encoder half: 106 tensors under `encoder.*` (8 resnets + 2 mid resnets: 10 x 8 + 2 shortcuts x 2,
    attention 10, 3 downsamplers x 2, conv_in/conv_out/conv_norm_out 6) + 2 under `quant_conv.*`."""
    enc = AutoencoderKLEncoder(device="meta")
    keys = list(enc.state_dict().keys())
    assert len([k for k in keys if k.startswith("encoder.")]) == 106 and len(keys) == 108
    assert enc.encoder.conv_in.weight.shape == (128, 3, 3, 3) and enc.encoder.conv_out.weight.shape == (8, 512, 3, 3)
    assert enc.encoder.down_blocks[1].resnets[0].conv_shortcut.weight.shape == (256, 128, 1, 1)
    assert enc.encoder.down_blocks[3].downsamplers is None and enc.quant_conv.weight.shape == (8, 8, 1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("b,f,hw", [(1, 2, (8, 8)), (2, 3, (16, 8))])
def test_decode_latents_gpu_parity(b, f, hw):
    """Real SD1.5 VAE widths (128/256/512/512, single-head 512-wide attention) at a small latent so the oracle runs in seconds.
    Tolerance: bf16 storage through ~60 dependent kernels: relative L2 <= 3e-2 (same bar as the UNet)."""
    ref = R.init_synthetic_weights(R.VAEDecoderRef(), seed=0).eval()
    vae = AutoencoderKLDecoder(device="cuda")
    vae.load_state_dict(ref.state_dict(), strict=True)
    vae = vae.to(torch.bfloat16).eval()
    lat = torch.randn(b, 4, f, *hw, generator=torch.Generator().manual_seed(2)) * 0.18215
    want = ref.decode_latents(lat)
    got = vae.decode_latents(lat.cuda())
    assert got.shape == want.shape and got.dtype == torch.float32 and torch.isfinite(got).all()
    rel = ((got.cpu() - want).norm() / want.norm()).item()
    print(f"[parity] VAE decode b={b} f={f} latent {hw}: rel_l2={rel:.3e} (|ref|max {want.abs().max().item():.3e})")
    assert rel <= 3e-2


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(64, 64), (64, 96)])
def test_encode_gpu_parity(hw):
    """``encode_latents`` (pipeline.py:540-562) on the HIP kernels at the real SD1.5 encoder widths: posterior moments against
    the oracle (latent 8 x 8 and 8 x 12: 64 / 96 mid-block tokens, the latter through the zero-padded P V contraction), and the
    sampled latents with a CPU generator (drawn on the generator's device, as diffusers does).  Bar 3e-2 (bf16 storage through
    ~45 dependent kernels; the log-variance head is the noisiest output)."""
    from animate3d_amd.vae import AutoencoderKLEncoder
    ref = R.init_synthetic_weights(R.VAEEncoderRef(), seed=1).eval()
    enc = AutoencoderKLEncoder(device="cuda")
    enc.load_state_dict(ref.state_dict(), strict=True)
    enc = enc.to(torch.bfloat16).eval()
    x = torch.rand(2, 3, *hw, generator=torch.Generator().manual_seed(3)) * 2 - 1
    mw, lw = ref.encode(x)
    mg, lg = enc.encode(x.cuda())
    rel_m = ((mg.cpu() - mw).norm() / mw.norm()).item()
    rel_l = ((lg.cpu() - lw).norm() / lw.norm()).item()
    print(f"[parity] VAE encode 2x3x{hw[0]}x{hw[1]}: rel_l2 mean {rel_m:.3e} logvar {rel_l:.3e}")
    assert mg.shape == (2, 4, hw[0] // 8, hw[1] // 8) and torch.isfinite(mg).all() and rel_m <= 3e-2 and rel_l <= 3e-2
    z = enc.encode_latents(x.cuda(), generator=torch.Generator().manual_seed(7))
    noise = torch.randn(mw.shape, generator=torch.Generator().manual_seed(7))
    want = (mw + torch.exp(0.5 * lw) * noise) * 0.18215
    rel_z = ((z.cpu() - want).norm() / want.norm()).item()
    print(f"[parity] VAE encode_latents with a CPU generator: rel_l2 {rel_z:.3e}")
    assert z.is_cuda and rel_z <= 3e-2


@pytest.mark.gpu
def test_vae_kernels_gpu():
    """The three entry points added for the VAE against plain torch."""
    from animate3d_amd.hip_ops import HipOps
    ops = HipOps()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(300, 512, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(264, 512, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    s = ops.gemm_f32out(x, w, alpha=0.125)
    assert s.dtype == torch.float32
    torch.testing.assert_close(s, 0.125 * (x.float() @ w.float().t()), rtol=1e-4, atol=1e-3)
    p = ops.softmax_rows(s)
    torch.testing.assert_close(p.float(), torch.softmax(s, -1), rtol=1e-2, atol=1e-5)
    z = torch.randn(5, 4, 7, 9, generator=g, device="cuda")
    wm, bm = torch.randn(4, 4, generator=g, device="cuda"), torch.randn(4, generator=g, device="cuda")
    torch.testing.assert_close(ops.channel_mix(z, wm, bm, 5.5), 5.5 * torch.einsum("oc,bchw->bohw", wm, z) + bm[None, :, None, None], rtol=1e-5, atol=1e-5)
    # GroupNorm with 4 channels per group (C = 128, 32 groups: the last decoder level)
    from tests.torch_ops import TorchRefOps
    ref = TorchRefOps(torch.float32, "cuda")
    xg = torch.randn(2 * 100, 128, generator=g, device="cuda").to(torch.bfloat16)
    ga, be = torch.rand(128, device="cuda") + 0.5, torch.randn(128, device="cuda") * 0.1
    got, want = ops.group_norm(xg, 2, 100, ga, be, 32, 1e-6, True).float(), ref.group_norm(xg, 2, 100, ga, be, 32, 1e-6, True).float()
    assert (got - want).abs().max().item() <= 3 * 2.0 ** -8 * want.abs().max().item()


def test_vae_decode_flop_model():
    """2.51 TFLOP per 64x64 latent (SD1.5 decoder; hand count in profiles/README.md); scales with the pixel count except for the
    L^2 attention term."""
    from animate3d_amd.flops import vae_decode_flops
    f64 = vae_decode_flops(64, 64)
    assert abs(f64 / 1e12 - 2.5145) < 1e-3
    attn64, attn32 = 2 * 2.0 * 4096 * 4096 * 512, 2 * 2.0 * 1024 * 1024 * 512
    assert abs((f64 - attn64) / 4 - (vae_decode_flops(32, 32) - attn32)) < 1e6
