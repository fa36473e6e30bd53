"""Generate tests/golden/sds.npz by running the REFERENCE's own 4D-SDS step.

Run in the build container only (needs /root/reference):

    python -B tests/golden/make_sds_goldens.py

``AnimateMVDiffusionGuidance.compute_mvdream_recon_loss`` and ``get_camera_cond``
(custom/threestudio-animate3d/guidance/animatemv_guidance.py:345-358, 391-507) and ``normalize_camera``
(animatediff/pipelines/pipeline.py:176-190) are compiled from the reference files as they lie (the module itself imports
threestudio / diffusers, absent here, so only these function definitions are executed) and called with: the stand-in UNet and
the restated DDIMScheduler methods of tests/sds_stub.py, a fixed text-embedding tensor for ``prompt_utils``.  What the vectors
pin is the reference-owned input assembly and epilogue: first-frame handling, per-view timestep expansion, the (text, uncond)
CFG order and ITS combine formula, camera normalisation, x0 reconstruction, std rescale, first-frame re-pin, loss scaling.

Only data (seeded inputs and the reference's outputs) is written; no reference source leaves /root/reference.
"""
import ast
import math
import os
import sys
from types import SimpleNamespace

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn.functional as F
from einops import rearrange

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.sds_stub import StubDDIM, stub_unet  # noqa: E402

REF = "/root/reference"


class _Ann:                       # jaxtyping-style annotations in the signatures: Float[Tensor, "B 4 32 32"]
    def __class_getitem__(cls, item):
        return cls


def _functions(path, names, in_class=None):
    tree = ast.parse(open(path).read())
    body = tree.body
    if in_class:
        body = next(nd for nd in tree.body if isinstance(nd, ast.ClassDef) and nd.name == in_class).body
    out = [nd for nd in body if isinstance(nd, ast.FunctionDef) and nd.name in names]
    for nd in out:
        nd.decorator_list = []
    assert len(out) == len(names), (names, [nd.name for nd in out])
    return out


def main():
    ns = {"torch": torch, "np": np, "math": math, "F": F, "rearrange": rearrange, "Float": _Ann, "Int": _Ann, "Tensor": _Ann,
          "PromptProcessorOutput": _Ann}
    exec(compile(ast.Module(body=_functions(os.path.join(REF, "animatediff/pipelines/pipeline.py"), ("normalize_camera",)), type_ignores=[]),
                 "pipeline_camera", "exec"), ns)
    exec(compile(ast.Module(body=_functions(os.path.join(REF, "custom/threestudio-animate3d/guidance/animatemv_guidance.py"),
                                            ("compute_mvdream_recon_loss", "get_camera_cond"), in_class="AnimateMVDiffusionGuidance"),
                            type_ignores=[]), "guidance_sds", "exec"), ns)
    out = {}
    cases = [("rescale", dict(n_view=2, n_frame=4, guidance_scale=7.5, recon_std_rescale=0.5, i2v_cond_time_zero=False), 431),
             ("plain", dict(n_view=4, n_frame=3, guidance_scale=100.0, recon_std_rescale=0.0, i2v_cond_time_zero=True), 57)]
    for tag, cfg, tval in cases:
        g = torch.Generator().manual_seed(len(tag))
        n, f, hw = cfg["n_view"], cfg["n_frame"], 8
        B = n * f                                                   # batch_size 1 (the reference's [b]-shaped alpha only broadcasts then)
        latents = torch.randn(B, 4, hw, hw, generator=g)
        t = torch.tensor([tval])
        text = torch.randn(2 * n, 5, 16, generator=g)               # (text, uncond) order, as prompt_utils returns it
        image_embeds = torch.randn(n, 12, generator=g)
        c2w = torch.eye(4).repeat(B, 1, 1) + 0.3 * torch.randn(B, 4, 4, generator=g)
        ang = torch.zeros(B)
        calls = []

        def forward_unet(latents, t, encoder_hidden_states, camera, i2v_cond_time_zero, added_cond_kwargs=None):
            calls.append(dict(sample=latents.clone(), t=t.clone(), camera=camera.clone(), image_embeds=added_cond_kwargs["image_embeds"].clone()))
            return stub_unet(latents, t, encoder_hidden_states=encoder_hidden_states, camera=camera, added_cond_kwargs=added_cond_kwargs,
                             i2v_cond_time_zero=i2v_cond_time_zero).sample

        me = SimpleNamespace(cfg=SimpleNamespace(camera_condition_type="rotation", view_dependent_prompting=False, **cfg),
                             scheduler=StubDDIM(), forward_unet=forward_unet)
        me.get_camera_cond = lambda camera, fovy=None: ns["get_camera_cond"](me, camera, fovy)
        prompt_utils = SimpleNamespace(use_perp_neg=False, get_text_embeddings=lambda *a, **k: text)
        noise_seen = []
        real_randn_like = torch.randn_like
        torch.manual_seed(1234)

        def randn_like(x, *a, **k):
            r = real_randn_like(x, *a, **k)
            noise_seen.append(r.clone())
            return r

        torch.randn_like = randn_like
        try:
            lat = latents.clone().requires_grad_(True)
            loss, aux = ns["compute_mvdream_recon_loss"](me, lat, t, prompt_utils, ang, ang, ang, c2w, image_embeds)
            loss.backward()
        finally:
            torch.randn_like = real_randn_like
        assert len(calls) == 1 and len(noise_seen) == 1
        for k, v in dict(latents=latents, t=t, text=text, image_embeds=image_embeds, c2w=c2w, noise=noise_seen[0], loss=loss.detach(),
                         grad=lat.grad, latents_noisy=aux["latents_noisy"], noise_pred=aux["noise_pred"], latents_recon=aux["latents_recon"],
                         unet_sample=calls[0]["sample"], unet_t=calls[0]["t"], unet_camera=calls[0]["camera"],
                         unet_image_embeds=calls[0]["image_embeds"]).items():
            out[f"{tag}/{k}"] = v.detach().numpy().copy()
        out[f"{tag}/cfg"] = np.array([cfg["n_view"], cfg["n_frame"], cfg["guidance_scale"], cfg["recon_std_rescale"], float(cfg["i2v_cond_time_zero"])])
    path = os.path.join(HERE, "sds.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
