"""GPU end-to-end checks of the training path (reference train.py:343-357, 540-601) at the real SD1.5 widths:

(1) parameter gradients of one loss evaluation — HIP forward + HIP backward behind ``loss.backward()`` — against torch autograd
    on the CPU oracle (fp32), next to an honest 16-bit execution of the SAME autograd graph on plain-torch ops (tests/torch_ops.py,
    rounding to the storage type at the same points): the kernels must not be noisier than that (<= 1.5 x its error) and stay
    under an absolute bar.  Storage fp16 (the reference's autocast type) is the sharp check, bf16 the default.
(2) ``training_step`` (noise, first frame clean, MSE, backward, clip, fused AdamW over the flat buffer) for two steps.
"""
import pytest
import torch
import torch.nn.functional as F

from animate3d_amd.config import UNetConfig
from animate3d_amd.train import FlatAdamW, select_trainable, training_step
from animate3d_amd.unet import MVUNetMotionModel
from oracle import unet_ref as O
from tests.conftest import oracle_unet
from tests.torch_ops import TorchRefOps

pytestmark = pytest.mark.gpu

N_VIEWS, FRAMES, HW = 2, 3, (16, 16)
BAR = {torch.bfloat16: 6e-2, torch.float16: 1e-2}          # relative L2 of the whole flattened gradient vs the fp32 oracle


def _cuda(inp):
    return {k: (v.cuda() if torch.is_tensor(v) else ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) else v)) for k, v in inp.items()}


@pytest.fixture(scope="module")
def oracle():
    ocfg = O.UNetConfig()
    ref = oracle_unet(ocfg, N_VIEWS, FRAMES, HW, seed=0).train()
    select_trainable(ref)
    inp = O.synthetic_inputs(ocfg, N_VIEWS, N_VIEWS, FRAMES, HW, seed=3, cfg_doubled=False)
    target = torch.randn(N_VIEWS, 4, FRAMES - 1, *HW, generator=torch.Generator().manual_seed(1))
    pred = type(ref).forward.__wrapped__(ref, **inp).sample          # the oracle's forward body is plain differentiable torch
    loss = F.mse_loss(pred[:, :, 1:], target)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in ref.named_parameters() if p.requires_grad}
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    return ocfg, sd, inp, target, float(loss), grads


def _run(sd, inp, target, dtype, ops=None):
    model = MVUNetMotionModel(UNetConfig(), num_views=N_VIEWS, device="cuda", ops=ops)
    model.load_state_dict(sd, strict=True)
    select_trainable(model)
    model.enable_training(compute_dtype=dtype if ops is None else None)
    pred = model(**_cuda(inp)).sample
    loss = F.mse_loss(pred[:, :, 1:].float(), target.cuda())
    # fp16 activations gradients underflow without loss scaling (the reference trains under a GradScaler, train.py:583-590)
    scale = 8192.0 if dtype == torch.float16 else 1.0
    (loss * scale).backward()
    return float(loss), {k: p.grad.detach().float().cpu() / scale for k, p in model.named_parameters() if p.requires_grad}


def _flat_err(got, want):
    num = sum(float((got[k] - want[k]).norm() ** 2) for k in want)
    den = sum(float(want[k].norm() ** 2) for k in want)
    per = sorted(float((got[k] - want[k]).norm() / (want[k].norm() + 1e-30)) for k in want if float(want[k].norm()) > 1e-6 * den ** 0.5)
    return (num / den) ** 0.5, per[len(per) // 2], per[-1]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_unet_parameter_gradients_vs_oracle(oracle, dtype):
    ocfg, sd, inp, target, loss_ref, g_ref = oracle
    loss_hip, g_hip = _run(sd, inp, target, dtype)
    loss_emu, g_emu = _run(sd, inp, target, dtype, ops=TorchRefOps(dtype, "cuda"))
    assert set(g_hip) == set(g_ref) and all(torch.isfinite(g).all() for g in g_hip.values())
    e_hip, e_emu = _flat_err(g_hip, g_ref), _flat_err(g_emu, g_ref)
    print(f"[parity] training loss: hip {loss_hip:.6f} emulation {loss_emu:.6f} oracle {loss_ref:.6f}")
    print(f"[parity] {len(g_ref)} trainable tensors, gradient vs fp32 oracle ({dtype}): hip rel_l2 {e_hip[0]:.3e} (per-tensor median {e_hip[1]:.3e}, "
          f"worst {e_hip[2]:.3e}); 16-bit emulation of the same graph {e_emu[0]:.3e} ({e_emu[1]:.3e}, {e_emu[2]:.3e})")
    assert abs(loss_hip - loss_ref) < (2e-2 if dtype == torch.bfloat16 else 3e-3) * loss_ref
    assert e_hip[0] <= BAR[dtype], e_hip
    assert e_hip[0] <= 1.5 * e_emu[0] + 1e-3, (e_hip, e_emu)
    frozen = [k for k, v in sd.items() if not any(t in k for t in ("i2v.", "motion_modules."))]
    assert len(frozen) > 0


def test_training_steps_run_on_the_gpu(oracle):
    ocfg, sd, inp, target, loss_ref, g_ref = oracle
    from animate3d_amd.denoise import ddim_schedule
    model = MVUNetMotionModel(UNetConfig(), num_views=N_VIEWS, device="cuda")
    model.load_state_dict(sd, strict=True)
    params = select_trainable(model)
    model.enable_training()
    opt = FlatAdamW(params, model.ops, lr=1e-4, max_grad_norm=1.0)
    g = torch.Generator(device="cuda").manual_seed(5)
    lat = torch.randn(1, N_VIEWS, 4, FRAMES, *HW, generator=g, device="cuda") * 0.5
    text = torch.randn(1, 77, 768, generator=g, device="cuda")
    cams = inp["camera"].cuda()
    img = torch.randn(N_VIEWS, 1024, generator=g, device="cuda")
    before = opt.flat_p.clone()
    infos = [training_step(model, opt, lat, text, cams, img, alphas_cumprod=ddim_schedule(25)[1], num_views=N_VIEWS, generator=g) for _ in range(2)]
    for i, info in enumerate(infos):
        print(f"[parity] GPU training step {i}: loss {info['loss']:.5f} grad_norm {info['grad_norm']:.4f} skipped {info['skipped']}")
        assert not info["skipped"] and info["loss"] == info["loss"] and info["grad_norm"] > 0
    moved = (opt.flat_p - before).abs()
    assert float(moved.max()) <= 2.5e-4 and float((moved > 0).float().mean()) > 0.9       # two AdamW steps of lr 1e-4
    with torch.no_grad():                                                                   # validation forward: a fresh inference pack
        y = model(**_cuda(inp)).sample
    assert torch.isfinite(y).all() and not y.requires_grad


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_gradient_checkpointing_on_the_hip_kernels(dtype):
    """``enable_gradient_checkpointing()`` (train.py:381-382) on the real kernels: layers are recomputed inside backward() through the
    same autograd functions (attention with forward statistics, the in-place IP-Adapter accumulation, the weight-gradient kernels): the
    loss is bit-identical, the parameter gradients agree to rounding (autograd sums the gradients of multiply-used 16-bit activations
    in a different order when a layer's backward is replayed from its boundary; on the fp32 CPU op set the same check is bit-exact,
    tests/test_training_host.py), and the peak memory of the step drops."""
    model = MVUNetMotionModel(UNetConfig(), num_views=N_VIEWS, device="cuda").init_synthetic(seed=4)
    select_trainable(model)
    model.enable_training(compute_dtype=dtype)
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(N_VIEWS, 4, FRAMES, *HW, generator=g, device="cuda")
    t = torch.full((N_VIEWS,), 400, device="cuda")
    text = torch.randn(N_VIEWS, 77, 768, generator=g, device="cuda")
    img = torch.randn(N_VIEWS, 1024, generator=g, device="cuda")
    from animate3d_amd.embeddings import get_camera
    cams = get_camera(N_VIEWS).cuda()
    target = torch.randn(N_VIEWS, 4, FRAMES, *HW, generator=g, device="cuda")

    def run():
        for p in model.parameters():
            p.grad = None
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = F.mse_loss(model(x, t, encoder_hidden_states=text, camera=cams, num_views=N_VIEWS, added_cond_kwargs={"image_embeds": img}).sample.float(), target)
        (loss * 256.0).backward()
        return float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters() if p.requires_grad}, torch.cuda.max_memory_allocated() - base

    loss0, g0, peak0 = run()
    model.enable_gradient_checkpointing()
    loss1, g1, peak1 = run()
    print(f"[parity] {dtype}: loss {loss0:.6f} / {loss1:.6f}; peak memory of the step {peak0 / 2**20:.0f} MiB -> {peak1 / 2**20:.0f} MiB with checkpointing")
    assert loss0 == loss1
    # merge weights are excluded from the per-tensor view: d mix_factor is a difference of large sums (autograd_ops._Gemm.backward), its
    # relative error is unbounded under ANY reordering of 16-bit additions; it is covered by the whole-gradient norm below
    per = sorted((float((g0[k].float() - g1[k].float()).norm() / (g0[k].float().norm() + 1e-30)), k) for k in g0 if not k.endswith("mix_factor"))
    num = sum(float((g0[k].float() - g1[k].float()).norm() ** 2) for k in g0) ** 0.5
    den = sum(float(g0[k].float().norm() ** 2) for k in g0) ** 0.5
    print(f"[parity] {dtype}: gradients with vs without checkpointing: rel L2 {num / den:.3e}, per tensor median {per[len(per) // 2][0]:.3e}, "
          f"worst {per[-1][0]:.3e} ({per[-1][1]})")
    # same bar as the kernels-vs-oracle check above: a replayed layer may re-order 16-bit gradient sums, it may not be noisier than the arithmetic itself
    assert num / den <= BAR[dtype] and per[len(per) // 2][0] <= BAR[dtype]
    assert peak1 < 0.8 * peak0
