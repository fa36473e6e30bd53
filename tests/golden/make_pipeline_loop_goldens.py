"""Generate tests/golden/pipeline_loop.npz by running the REFERENCE's own denoising loop.

Run in the build container only (needs /root/reference):

    python -B tests/golden/make_pipeline_loop_goldens.py

The statement ``for free_init_iter in range(num_free_init_iters): ...`` of ``AnimateDiffMVI2VPipeline.__call__``
(animatediff/pipelines/pipeline.py:988-1045: the FreeInit passes, step 8's loop with the CFG combine, ``scheduler.step`` and the
first-frame re-pin) is taken from the reference file's syntax tree as it lies and compiled as the body of a function whose
parameters are the names it reads (the module itself imports diffusers / torchvision, absent here).  It runs with: the stand-in
UNet of tests/sds_stub.py, and restatements of the third-party pieces it calls — diffusers' ``DDIMScheduler`` (timesteps,
``scale_model_input`` = identity, ``step``) and ``FreeInitMixin._apply_free_init`` from oracle/denoise_ref.py.  What the vectors
pin is the reference-owned loop: which frames FreeInit touches, the (uncond, text) order and combine, the camera doubling, the
re-pin of the conditioning frame after every step.

Only data (seeded inputs and the reference's outputs) is written; no reference source leaves /root/reference.
"""
import ast
import contextlib
import os
import sys
from types import SimpleNamespace

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.denoise_ref import DDIMRef, FreeInitRef  # noqa: E402
from tests.sds_stub import stub_unet  # noqa: E402

REF = "/root/reference"
PARAMS = ["self", "latents", "first_frame_latents", "timesteps", "num_inference_steps", "device", "generator", "i2v_similarity_init",
          "strength", "prompt_embeds", "negative_prompt_embeds", "camera", "cross_attention_kwargs", "added_cond_kwargs",
          "i2v_cond_time_zero", "guidance_scale", "extra_step_kwargs", "callback_on_step_end", "callback_on_step_end_tensor_inputs",
          "callback", "callback_steps", "num_free_init_iters"]


def reference_loop():
    tree = ast.parse(open(os.path.join(REF, "animatediff/pipelines/pipeline.py")).read())
    cls = next(nd for nd in tree.body if isinstance(nd, ast.ClassDef) and nd.name == "AnimateDiffMVI2VPipeline")
    call = next(nd for nd in cls.body if isinstance(nd, ast.FunctionDef) and nd.name == "__call__")
    loop = next(nd for nd in ast.walk(call) if isinstance(nd, ast.For) and isinstance(nd.target, ast.Name) and nd.target.id == "free_init_iter")
    fn = ast.FunctionDef(name="ref_loop", args=ast.arguments(posonlyargs=[], args=[ast.arg(arg=a) for a in PARAMS], kwonlyargs=[],
                                                             kw_defaults=[], defaults=[]),
                         body=[loop, ast.Return(value=ast.Name(id="latents", ctx=ast.Load()))], decorator_list=[])
    mod = ast.fix_missing_locations(ast.Module(body=[fn], type_ignores=[]))
    ns = {"torch": torch}
    exec(compile(mod, "pipeline_loop", "exec"), ns)
    return ns["ref_loop"]


class Scheduler(DDIMRef):
    order = 1

    def scale_model_input(self, sample, t):
        return sample

    def step(self, eps, t, sample, **kw):
        return SimpleNamespace(prev_sample=DDIMRef.step(self, eps, int(t), sample))


def main():
    ref_loop = reference_loop()
    out = {}
    for tag, free_init, steps, n, F, hw, scale, tz in [("freeinit", True, 4, 2, 4, 8, 7.5, False), ("plain", False, 5, 4, 3, 8, 3.0, True)]:
        g = torch.Generator().manual_seed(len(tag))
        first = 0.18215 * torch.randn(n, 4, 1, hw, hw, generator=g)
        latents = torch.cat([first, torch.randn(n, 4, F - 1, hw, hw, generator=g)], dim=2)
        prompt = torch.randn(2 * n, 5, 16, generator=g)                # (uncond, text)
        embeds = torch.randn(2 * n, 12, generator=g)
        camera = torch.randn(n, 16, generator=g)
        sched = Scheduler()
        sched.set_timesteps(steps)
        fi = FreeInitRef(sched)
        unet_calls = []

        def unet(sample, t, **kw):
            unet_calls.append((sample.clone(), int(t), kw["camera"].clone()))
            kw.pop("cross_attention_kwargs")
            return stub_unet(sample, t, **kw)

        me = SimpleNamespace(free_init_enabled=free_init, _free_init_num_iters=3, scheduler=sched, unet=unet, do_classifier_free_guidance=True,
                             progress_bar=lambda total: contextlib.nullcontext(SimpleNamespace(update=lambda: None)),
                             _apply_free_init=lambda lat, it, nsteps, device, dtype, generator: (fi.apply(lat, it, generator), sched.timesteps))
        gen = torch.Generator().manual_seed(77)
        res = ref_loop(me, latents.clone(), first, sched.timesteps, steps, "cpu", gen, None, 1.0, prompt, None, camera, None,
                       {"image_embeds": embeds}, tz, scale, {}, None, [], None, 1, 3 if free_init else 1)
        for k, v in dict(latents=latents, first=first, prompt=prompt, embeds=embeds, camera=camera, result=res,
                         first_unet_sample=unet_calls[0][0], last_unet_sample=unet_calls[-1][0], unet_camera=unet_calls[0][2]).items():
            out[f"{tag}/{k}"] = v.numpy().copy()
        out[f"{tag}/cfg"] = np.array([float(free_init), steps, n, F, scale, float(tz), len(unet_calls)])
        out[f"{tag}/unet_t"] = np.array([c[1] for c in unet_calls])
    # ---- step 5, ``prepare_latents`` (pipeline.py:677-733) with ``i2v_similarity_init`` = None, and the first-frame concat (:950-973)
    tree = ast.parse(open(os.path.join(REF, "animatediff/pipelines/pipeline.py")).read())
    cls = next(nd for nd in tree.body if isinstance(nd, ast.ClassDef) and nd.name == "AnimateDiffMVI2VPipeline")
    prep = next(nd for nd in cls.body if isinstance(nd, ast.FunctionDef) and nd.name == "prepare_latents")
    ns = {"torch": torch, "randn_tensor": lambda shape, generator=None, device=None, dtype=None: torch.randn(shape, generator=generator, dtype=dtype)}
    exec(compile(ast.fix_missing_locations(ast.Module(body=[prep], type_ignores=[])), "prepare_latents", "exec"), ns)
    me = SimpleNamespace(vae_scale_factor=8, scheduler=SimpleNamespace(init_noise_sigma=1.0))
    first = torch.randn(3, 4, 8, 8, generator=torch.Generator().manual_seed(5)).unsqueeze(2)
    rest = ns["prepare_latents"](me, 3, 4, 6 - 1, 64, 64, torch.float32, "cpu", torch.Generator().manual_seed(21), None, None,
                                 latent_timestep=None, latent_cond_image=first)
    out["prepare/first"] = first.numpy().copy()
    out["prepare/latents"] = torch.cat([first, rest], dim=2).numpy().copy()
    path = os.path.join(HERE, "pipeline_loop.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
