"""The SDS-side input assembly and epilogue (SURVEY.md §8c "Python caller rows") against golden vectors produced by the
REFERENCE's own ``compute_mvdream_recon_loss`` (tests/golden/make_sds_goldens.py), with the stand-in UNet of tests/sds_stub.py on
both sides."""
import os

import numpy as np
import pytest
import torch

from animate3d_amd.sds import normalize_camera, sds_recon_loss
from tests.sds_stub import StubDDIM, stub_unet

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sds.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLD)


@pytest.mark.parametrize("tag", ["rescale", "plain"])
def test_sds_step_matches_the_reference(golden, tag):
    G = {k.split("/", 1)[1]: torch.from_numpy(golden[k]) for k in golden.files if k.startswith(tag + "/")}
    n, f, scale, rescale, tz = G["cfg"].tolist()
    seen = {}

    def unet(sample, timestep, **kw):
        seen.update(sample=sample.clone(), t=timestep.clone(), camera=kw["camera"].clone(), image_embeds=kw["added_cond_kwargs"]["image_embeds"].clone())
        return stub_unet(sample, timestep, **kw)

    lat = G["latents"].clone().requires_grad_(True)
    loss, aux = sds_recon_loss(unet, lat, G["t"], G["text"], G["image_embeds"], G["c2w"], n_view=int(n), n_frame=int(f), guidance_scale=scale,
                               recon_std_rescale=rescale, i2v_cond_time_zero=bool(tz), alphas_cumprod=StubDDIM().alphas_cumprod, noise=G["noise"])
    loss.backward()
    # what the UNet is handed: CFG-doubled noisy videos with the clean first frame, per-view timesteps, normalised cameras of
    # frame 0, image embeddings followed by zeros
    torch.testing.assert_close(seen["sample"], G["unet_sample"], rtol=1e-6, atol=1e-6)
    assert torch.equal(seen["t"].long(), G["unet_t"].long())
    torch.testing.assert_close(seen["camera"], G["unet_camera"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(seen["image_embeds"], G["unet_image_embeds"], rtol=0, atol=0)
    torch.testing.assert_close(aux["latents_noisy"], G["latents_noisy"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(aux["noise_pred"], G["noise_pred"], rtol=1e-5, atol=1e-4)          # guidance scale 100 amplifies rounding
    torch.testing.assert_close(aux["latents_recon"], G["latents_recon"], rtol=1e-5, atol=2e-4)
    torch.testing.assert_close(loss.detach(), G["loss"], rtol=1e-5, atol=0)
    torch.testing.assert_close(lat.grad, G["grad"], rtol=1e-5, atol=1e-5)
    # structural: frame 0 of the target is the input's frame 0, so it carries no gradient
    g6 = lat.grad.reshape(-1, int(f), *lat.shape[1:])
    assert float(g6[:, 0].abs().max()) == 0.0 and float(g6[:, 1:].abs().max()) > 0.0


def test_sds_batches_and_argument_checks():
    """b = 2 equals two independent b = 1 steps (alpha per b — the reference's [b]-shaped broadcast only exists for b = 1),
    own noise from a generator, and the error behaviour."""
    n, f, hw = 2, 3, 4
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(2 * n * f, 4, hw, hw, generator=g)
    t = torch.tensor([100, 700])
    text = torch.randn(2 * 2 * n, 5, 16, generator=g)               # (text for b0 b1, uncond for b0 b1)
    emb = torch.randn(2 * n, 12, generator=g)
    c2w = torch.eye(4).repeat(2 * n * f, 1, 1) + 0.2 * torch.randn(2 * n * f, 4, 4, generator=g)
    noise = torch.randn(2, n, f - 1, 4, hw, hw, generator=g)
    kw = dict(n_view=n, n_frame=f, guidance_scale=7.5, recon_std_rescale=0.5)
    loss, aux = sds_recon_loss(stub_unet, lat, t, text, emb, c2w, noise=noise, **kw)
    parts = []
    for i in range(2):
        rows = slice(i * n * f, (i + 1) * n * f)
        txt = torch.cat([text[i * n:(i + 1) * n], text[2 * n + i * n: 2 * n + (i + 1) * n]])
        li, ai = sds_recon_loss(stub_unet, lat[rows], t[i:i + 1], txt, emb[i * n:(i + 1) * n], c2w[rows], noise=noise[i:i + 1], **kw)
        parts.append((li, ai))
        torch.testing.assert_close(aux["latents_recon"][rows], ai["latents_recon"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(loss, (parts[0][0] + parts[1][0]) / 2, rtol=1e-5, atol=0)
    l1, _ = sds_recon_loss(stub_unet, lat, t, text, emb, c2w, generator=torch.Generator().manual_seed(3), **kw)
    l2, _ = sds_recon_loss(stub_unet, lat, t, text, emb, c2w, generator=torch.Generator().manual_seed(3), **kw)
    assert torch.equal(l1, l2) and torch.isfinite(l1)
    cam = normalize_camera(c2w)
    torch.testing.assert_close(cam.reshape(-1, 4, 4)[:, :3, 3].norm(dim=1), torch.ones(2 * n * f), rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        sds_recon_loss(stub_unet, lat[:5], t, text, emb, c2w, **kw)
    with pytest.raises(ValueError):
        sds_recon_loss(stub_unet, lat, t[:1], text, emb, c2w, **kw)
    with pytest.raises(ValueError):
        sds_recon_loss(stub_unet, lat, t, text[:3], emb, c2w, **kw)


@pytest.mark.gpu
def test_sds_step_config5_over_the_hip_unet():
    """BASELINE config 5: one 4D-SDS step (animatemv_guidance.py:391-507) over the HIP UNet at the reference's shape — b = 1,
    4 views x 16 frames, 32 x 32 latent (256 px), everything cast to fp16 on the way in (:339-346), CFG batch in (text, uncond)
    order — against the same function over the fp32 CPU oracle UNet with the same weights, noise and timestep.  The oracle side
    (two CFG halves of a 0.66 PFLOP forward: 5.4 minutes of the GPU box's CPU) is computed once by
    tests/golden/make_gpu_tier_goldens.py and committed (tests/golden/gpu_tier_oracle.npz, float16); weights and inputs are re-drawn
    here from the same seeds through that script's own functions.
    The ``.half()`` model runs on the fp16-storage kernels (a3d_*_f16), as the reference runs its UNet in fp16.
    Compared: the raw UNet output (bar 6e-3, the fp16 tolerance of tests/test_unet_gpu.py; the golden's storage adds 3e-4) and, at
    guidance scale 7.5, the reconstruction, the loss and the gradient (the CFG combine eps_text + s (eps_text - eps_uncond)
    amplifies the UNet's rounding by ~ s, hence the wider bar 2.5e-2; at the reference's s = 100 only finiteness is asserted)."""
    from animate3d_amd.config import UNetConfig
    from animate3d_amd.unet import MVUNetMotionModel
    from tests.golden import make_gpu_tier_goldens as G
    gold = np.load(G.OUT)
    n, F, hw, b = G.SDS5["n"], G.SDS5["F"], G.SDS5["hw"], G.SDS5["b"]
    ref = G.sds5_weights()
    hip = MVUNetMotionModel(UNetConfig(), num_views=n, device="cuda")
    hip.load_state_dict(ref.state_dict(), strict=True)
    del ref
    hip = hip.half().eval()
    lat, noise, t, text, emb, c2w = G.sds5_inputs()
    assert abs(lat.double().sum().item() + text.double().sum().item() - float(gold["sds5_in_checksum"])) < 1e-6, "inputs differ from the golden's"
    seen = {}

    def unet(*a, **k):
        out = hip(*a, **k)
        seen["hip"] = out.sample.detach().float().cpu()
        return out

    kw = dict(n_view=n, n_frame=F, recon_std_rescale=G.SDS5["recon_std_rescale"])
    lh = lat.clone().cuda().requires_grad_(True)
    dev = lambda v: v.cuda()
    loss_h, aux_h = sds_recon_loss(unet, lh, dev(t), dev(text), dev(emb), dev(c2w), guidance_scale=G.SDS5["guidance_scale"],
                                   weights_dtype=torch.float16, noise=dev(noise), **kw)
    loss_h.backward()
    rel = lambda a, b_: ((a.float().cpu() - b_.float().cpu()).norm() / b_.float().cpu().norm()).item()
    want = lambda k: torch.from_numpy(gold[k].astype(np.float32))
    e_unet = rel(seen["hip"], want("sds5_unet"))
    e_rec = rel(aux_h["latents_recon"], want("sds5_recon"))
    e_grad = rel(lh.grad * float(gold["sds5_grad_scale"]), want("sds5_grad_scaled"))
    loss_r = float(gold["sds5_loss"])
    e_loss = abs(loss_h.item() - loss_r) / abs(loss_r)
    print(f"[parity] SDS step config 5 (V=8, F=16, 32x32, fp16 in) vs the oracle golden: UNet output rel_l2={e_unet:.3e}; s=7.5: recon {e_rec:.3e} "
          f"grad {e_grad:.3e} loss {e_loss:.3e} (loss {loss_r:.4e})")
    assert hip.ops.act_dtype == torch.float16
    assert seen["hip"].shape == (2 * b * n, 4, F, *hw) and e_unet <= 6e-3
    assert e_rec <= 2.5e-2 and e_grad <= 2.5e-2 and e_loss <= 1e-2
    g6 = lh.grad.reshape(-1, F, *lh.shape[1:])
    assert float(g6[:, 0].abs().max()) == 0.0
    loss_100, aux_100 = sds_recon_loss(hip, lat.cuda(), dev(t), dev(text), dev(emb), dev(c2w), guidance_scale=100.0,
                                       weights_dtype=torch.float16, noise=dev(noise), **kw)
    assert torch.isfinite(loss_100) and torch.isfinite(aux_100["latents_recon"]).all()
