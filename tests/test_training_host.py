"""CPU tests of the training path's HOST logic: the product UNet runs a grad-enabled forward on
``AutogradOps`` over the plain-torch op set of tests/torch_ops.py, and the gradients that ``loss.backward()`` leaves on its
parameters must equal those torch autograd leaves on the oracle's restatement of the reference forward for the loss of
train.py:540-577 (first frame clean, noise prediction on the rest).  This pins every backward composition of
animate3d_amd/autograd_ops.py (dgrad / wgrad through transposed operands, flipped conv weights, zero-stuffed stride-2 dgrad,
up-sampler backward, recomputed GEGLU projection, first-frame K/V sharing, accumulated IP-adapter attention, merge-weight
gradient) and the per-step differentiable weight packing of unet.py; the kernels themselves are -m gpu."""
import pytest
import torch
import torch.nn.functional as F

from animate3d_amd.config import UNetConfig
from animate3d_amd.unet import MVUNetMotionModel
from oracle import unet_ref as O
from tests.torch_ops import TorchRefOps

SMALL = dict(block_out_channels=(32, 64, 64, 64))
TRAINABLE = ("i2v.", "motion_modules.")          # configs/training/train.yaml: trainable_modules


def _pair(n, Fr, hw, **cfgkw):
    ocfg = O.UNetConfig(**SMALL, **cfgkw)
    ref = O.MVUNetMotionModelRef(ocfg, n, Fr, hw)
    O.init_synthetic_weights(ref, seed=0, dense=True)
    model = MVUNetMotionModel(UNetConfig(**SMALL, **cfgkw), ops=TorchRefOps(), num_views=n)
    model.load_state_dict(ref.state_dict(), strict=True)
    for m in (ref, model):                        # train.py:343-349
        m.requires_grad_(False)
        for name, p in m.named_parameters():
            if any(t in name for t in TRAINABLE):
                p.requires_grad = True
    return ocfg, ref, model


def _loss(unet, inp, target):
    # the oracle's forward is wrapped in no_grad (it restates the inference call sites); its body is plain differentiable torch
    fwd = type(unet).forward.__wrapped__.__get__(unet) if isinstance(unet, O.MVUNetMotionModelRef) else unet
    pred = fwd(**inp).sample                      # train.py:570-577
    return F.mse_loss(pred[:, :, 1:].float(), target.float(), reduction="mean")


@pytest.mark.parametrize("n,Fr,hw,kw", [
    (2, 3, (8, 8), {}),
    (2, 2, (12, 20), {}),                                                   # forced up-sample sizes in the backward
    (2, 2, (8, 8), dict(motion_use_alpha_blender=False)),
    (2, 2, (8, 8), dict(mvdream_image_attn=False, motion_spatial_attn=False)),
    # 16x16: with these switch sets the 1x1 deepest level of an 8x8 latent makes the fp32 gradient itself chaotic (the oracle's
    # own fp32 autograd is then 3-7 % off its float64 run on whole blocks)
    (2, 2, (16, 16), dict(motion_image_attn=True)),                        # first-frame image branch + 3-way SoftmaxAlphaBlender weights
    (2, 2, (16, 16), dict(motion_use_camera_encoding=True, motion_camera_encoding_type="sinusoid")),
])
def test_parameter_gradients_match_autograd_of_the_oracle(n, Fr, hw, kw):
    ocfg, ref, model = _pair(n, Fr, hw, **kw)
    inp = O.synthetic_inputs(ocfg, n, n, Fr, hw, seed=3, cfg_doubled=False)
    g = torch.Generator().manual_seed(1)
    target = torch.randn(n, 4, Fr - 1, hw[0], hw[1], generator=g)
    model.enable_training()
    # Yardstick = the oracle in float64.  Its own fp32 autograd is the noise floor: the gradients of the merge weights and of the
    # deepest layers are long, heavily cancelling sums (fp32 autograd of the oracle is off by up to ~5 % on a mix_factor), so the
    # bar per tensor is "within 5x the oracle's fp32-vs-fp64 difference" (or 1e-3 of the tensor's largest gradient).
    import copy
    ref64 = copy.deepcopy(ref).double()
    dbl = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else ({k: w.double() for k, w in v.items()} if isinstance(v, dict) else v)
    l_ref = _loss(ref64, {k: dbl(v) for k, v in inp.items()}, target.double())
    l_ref.backward()
    _loss(ref, inp, target).backward()
    l = _loss(model, inp, target)
    assert l.requires_grad
    l.backward()
    assert abs(l.item() - l_ref.item()) < 1e-4 * abs(l_ref.item())
    g64 = {k: p.grad for k, p in ref64.named_parameters() if p.requires_grad}
    g32 = {k: p.grad for k, p in ref.named_parameters() if p.requires_grad}
    got = {k: p.grad for k, p in model.named_parameters() if p.requires_grad}
    assert set(g64) == set(got) and len(got) > 100
    top = max(float(g.abs().max()) for g in g64.values() if g is not None)
    worst = 0.0
    for k, gr in g64.items():
        assert got[k] is not None, f"no gradient reached {k}"
        if gr is None:
            continue
        scale = float(gr.abs().max())
        err = float((got[k].double() - gr).abs().max())
        floor = float((g32[k].double() - gr).abs().max())
        rel = 2e-2 if k.endswith("mix_factor") else 1e-3      # merge weight: sum over W * dW (weight space) instead of dY * T (token space)
        assert err <= max(5.0 * floor, rel * scale, 1e-6 * top), (k, err, floor, scale)
        if scale > 1e-4 * top:
            worst = max(worst, err / scale)
    print(f"[parity] {len(got)} trainable tensors, worst max-abs gradient error / max |grad| (tensors above 1e-4 of the largest) = {worst:.2e}")
    # frozen parameters stay without gradient, and a no_grad forward is the inference path
    assert all(p.grad is None for k, p in model.named_parameters() if not p.requires_grad)
    with torch.no_grad():
        y = model(**inp).sample
    assert not y.requires_grad


def test_training_rejects_what_it_cannot_differentiate():
    ocfg, ref, model = _pair(2, 2, (8, 8))
    inp = O.synthetic_inputs(ocfg, 2, 2, 2, (8, 8), seed=3, cfg_doubled=False)
    model.enable_training()
    model.conv_in.weight.requires_grad = True
    with pytest.raises(NotImplementedError):
        model(**inp)
    model.conv_in.weight.requires_grad = False
    model.down_blocks[0].resnets[0].conv1.weight.requires_grad = True        # a 3x3 conv weight: no wgrad kernel
    with pytest.raises(NotImplementedError):
        model(**inp).sample.sum().backward()
    # learnable positional tables would be trainable by name ('motion_modules.'): refused loudly rather than left without a gradient
    ocfg, ref, model = _pair(2, 2, (8, 8), motion_spatial_encoding_type="learnable")
    model.enable_training()
    with pytest.raises(NotImplementedError):
        model(**O.synthetic_inputs(ocfg, 2, 2, 2, (8, 8), seed=3, cfg_doubled=False))
    for k, p in model.named_parameters():
        if "spatial_pos_embed" in k:
            p.requires_grad = False
    assert model(**O.synthetic_inputs(ocfg, 2, 2, 2, (8, 8), seed=3, cfg_doubled=False)).sample.requires_grad


# ------------------------------------------------------------------ the optimisation step (animate3d_amd/train.py)
def _batch(ocfg, b, n, Fr, hw, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(latents=torch.randn(b, n, 4, Fr, hw[0], hw[1], generator=g) * 0.5,
                text=torch.randn(b, 77, ocfg.cross_attention_dim, generator=g),
                cameras=torch.randn(b * n, 16, generator=g),
                image_embeds=torch.randn(b * n, ocfg.ip_image_embed_dim, generator=g),
                noise=torch.randn(b, n, 4, Fr - 1, hw[0], hw[1], generator=g),
                timesteps=torch.randint(0, 1000, (b,), generator=g))


def _reference_steps(ref, batches, n, steps_lr=1e-3):
    """train.py:351-357, 540-596 with stock torch on the oracle: AdamW + clip_grad_norm_(1.0), no autocast (fp32)."""
    from animate3d_amd.denoise import ddim_schedule
    from animate3d_amd.train import add_noise
    ac = ddim_schedule(25)[1]
    fwd = type(ref).forward.__wrapped__.__get__(ref)
    params = [p for p in ref.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=steps_lr, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    losses = []
    for bt in batches:
        lat = bt["latents"]
        b, nn_, c, f, h, w = lat.shape
        noisy = torch.cat([lat[:, :, :, :1], add_noise(lat[:, :, :, 1:], bt["noise"], bt["timesteps"], ac)], dim=3).reshape(b * nn_, c, f, h, w)
        ehs = bt["text"][:, None].expand(b, nn_, 77, -1).reshape(b * nn_, 77, -1)
        t = bt["timesteps"][:, None].expand(b, nn_).reshape(-1)
        added = None if bt["image_embeds"] is None else {"image_embeds": bt["image_embeds"]}
        pred = fwd(noisy, t, encoder_hidden_states=ehs, camera=bt["cameras"], num_views=n, added_cond_kwargs=added).sample
        loss = F.mse_loss(pred.reshape(b, nn_, c, f, h, w)[:, :, :, 1:].float(), bt["noise"].float())
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step()
        losses.append(float(loss))
    return losses


def _product_steps(model, batches, n, lr=1e-3, group=None, **optkw):
    from animate3d_amd.denoise import ddim_schedule
    from animate3d_amd.train import FlatAdamW, training_step
    ac = ddim_schedule(25)[1]
    model.enable_training()
    opt = FlatAdamW([p for p in model.parameters() if p.requires_grad], model.ops, lr=lr, **optkw)
    out = []
    for bt in batches:
        out.append(training_step(model, opt, bt["latents"], bt["text"], bt["cameras"], bt["image_embeds"], alphas_cumprod=ac, num_views=n,
                                 noise=bt["noise"], timesteps=bt["timesteps"], group=group))
    return out, opt


def test_training_steps_follow_torch_adamw_on_the_oracle():
    n, Fr, hw = 2, 3, (8, 8)
    ocfg, ref, model = _pair(n, Fr, hw)
    batches = [_batch(ocfg, 1, n, Fr, hw, seed=10 + i) for i in range(3)]
    p0 = {k: p.detach().clone() for k, p in ref.named_parameters() if p.requires_grad}
    want = _reference_steps(ref, batches[:1], n)
    g0 = {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.requires_grad}       # clipped gradient of step 0
    got, opt = _product_steps(model, batches[:1], n)
    assert abs(got[0]["loss"] - want[0]) < 1e-5 * abs(want[0]) and not got[0]["skipped"]
    # First AdamW step: every element moves by ~lr * sign(g) (plus weight decay).  Elements whose gradient is fp32 noise (e.g. a bias
    # in front of a GroupNorm: exactly zero in exact arithmetic) get a coin-flip sign in ANY implementation, so the comparison
    # runs over the elements that carry a gradient worth the name.
    pm, pr = dict(model.named_parameters()), dict(ref.named_parameters())
    tot = bad = 0
    for k, g in g0.items():
        sig = g.abs() > 1e-2 * g.abs().max().clamp_min(1e-30)
        d_ref, d_got = (pr[k].detach() - p0[k])[sig], (pm[k].detach() - p0[k])[sig]
        tot += int(sig.sum())
        bad += int(((d_ref - d_got).abs() > 0.05 * d_ref.abs()).sum())
    print(f"[parity] first AdamW step: {tot} parameter elements with a significant gradient, {bad} moved differently (> 5 %)")
    assert tot > 10_000 and bad <= 1e-3 * tot
    # two more steps from each side's own parameters: the losses stay together
    want += _reference_steps(ref, batches[1:], n)
    from animate3d_amd.denoise import ddim_schedule
    from animate3d_amd.train import training_step
    for bt in batches[1:]:
        got.append(training_step(model, opt, bt["latents"], bt["text"], bt["cameras"], bt["image_embeds"], alphas_cumprod=ddim_schedule(25)[1],
                                 num_views=n, noise=bt["noise"], timesteps=bt["timesteps"]))
    for i, (g, w) in enumerate(zip(got, want)):
        print(f"[parity] train step {i}: loss {g['loss']:.6f} (torch on the oracle {w:.6f}), grad norm {g['grad_norm']:.4f}")
    assert abs(got[1]["loss"] - want[1]) < 2e-3 * abs(want[1])
    assert all(p.data_ptr() >= opt.flat_p.data_ptr() for p in opt.params)


def test_loss_scaler_skips_and_backs_off_on_overflow():
    n, Fr, hw = 2, 2, (8, 8)
    ocfg, ref, model = _pair(n, Fr, hw)
    bt = _batch(ocfg, 1, n, Fr, hw, seed=3)
    bad = dict(bt, latents=bt["latents"].clone())
    bad["latents"][0, 0, 0, 1, 0, 0] = float("inf")
    before = {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
    got, opt = _product_steps(model, [bad, bt], n, loss_scale=1024.0)
    assert got[0]["skipped"] and opt.loss_scale == 512.0 and not got[1]["skipped"] and opt.step_count == 1
    changed = sum(int(not torch.equal(p, before[k])) for k, p in model.named_parameters() if p.requires_grad)
    assert changed > 100


# ------------------------------------------------------------------ data parallel (train.py: DDP) over gloo, world 2
def _dp_worker(rank, world, port, q):
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, Fr, hw = 2, 2, (8, 8)
        ocfg, ref, model = _pair(n, Fr, hw)
        from animate3d_amd.denoise import ddim_schedule
        from animate3d_amd.train import FlatAdamW, training_step
        bt = _batch(ocfg, 1, n, Fr, hw, seed=20 + rank)                      # this rank's shard of the global batch
        model.enable_training()
        opt = FlatAdamW([p for p in model.parameters() if p.requires_grad], model.ops, lr=1e-3, bucket_bytes=1 << 20)   # several buckets in flight
        seen = {}
        reduce = opt.all_reduce_grads

        def spy(group=None):                                                 # local gradient just before the exchange
            seen["local"] = opt.flat_g.clone()
            return reduce(group)
        opt.all_reduce_grads = spy
        info = training_step(model, opt, bt["latents"], bt["text"], bt["cameras"], bt["image_embeds"], alphas_cumprod=ddim_schedule(25)[1],
                             num_views=n, noise=bt["noise"], timesteps=bt["timesteps"])
        locals_ = [torch.zeros_like(opt.flat_g) for _ in range(world)]
        dist.all_gather(locals_, seen["local"])
        total = sum(locals_)
        err = float((opt.flat_g - total).abs().max() / total.abs().max())
        differ = float((locals_[0] - locals_[1]).abs().max() / total.abs().max())          # the shards really are different data
        norm = float((total / world)[:opt.numel].norm())
        gathered = [torch.zeros_like(opt.flat_p) for _ in range(world)]
        dist.all_gather(gathered, opt.flat_p)
        same = all(torch.equal(gathered[0], g) for g in gathered) and differ > 1e-2
        q.put((rank, err, same, abs(info["grad_norm"] - norm) / norm))
    except Exception as e:
        q.put((rank, repr(e), False, 0.0))
        raise
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradients_over_gloo():
    """One process per GPU, batch sharded (train.py:392-399 DistributedSampler): after the bucketed all-reduce of the flat gradient
    every rank holds the sum of the ranks' gradients, clips by the norm of their mean, takes the same AdamW step and stays
    bit-identical to its peers.  (That the mean of per-shard gradients is the gradient of the global batch's mean loss is
    arithmetic, not something this code decides; on these random fp32 networks it only holds to a few per cent numerically.)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, same, dnorm in res:
        assert not isinstance(err, str), err
        print(f"[parity] rank {rank}: flat gradient after the exchange vs sum of the ranks' gradients: max err / max |g| = {err:.2e}, "
              f"clip norm vs |mean gradient| {dnorm:.2e}")
        assert err < 1e-6 and same and dnorm < 1e-3          # fp32 sums over 2M elements in two orders


def test_validation_forward_after_each_optimiser_step_sees_the_updated_weights():
    """train.py:644-700 samples with the SAME unet between optimisation steps: the no_grad path must not reuse kernel-layout copies of
    parameters an optimiser step has changed since (and the next training step must not reuse the validation pack either)."""
    n, Fr, hw = 2, 2, (8, 8)
    ocfg, ref, model = _pair(n, Fr, hw)
    inp = O.synthetic_inputs(ocfg, n, n, Fr, hw, seed=3, cfg_doubled=False)
    batches = [_batch(ocfg, 1, n, Fr, hw, seed=30 + i) for i in range(2)]
    model.enable_gradient_checkpointing()                                  # train.py:381-382: accepted, keeps activations
    from animate3d_amd.denoise import ddim_schedule
    from animate3d_amd.train import FlatAdamW, training_step
    model.enable_training()
    opt = FlatAdamW([p for p in model.parameters() if p.requires_grad], model.ops, lr=1e-2)      # large steps: stale weights would show
    with torch.no_grad():
        y0 = model(**inp).sample
    for bt in batches:
        training_step(model, opt, bt["latents"], bt["text"], bt["cameras"], bt["image_embeds"], alphas_cumprod=ddim_schedule(25)[1],
                      num_views=n, noise=bt["noise"], timesteps=bt["timesteps"])
        with torch.no_grad():
            y = model(**inp).sample
        ref.load_state_dict(model.state_dict())                              # the oracle with the CURRENT weights
        want = ref(**inp).sample
        assert float((y - want).abs().max()) < 2e-3 * float(want.abs().max())
        assert float((y - y0).abs().max()) > 1e-2 * float(want.abs().max())    # and the step did change the function
        y0 = y


def test_flat_adamw_checkpoint_round_trip_and_grad_relinking():
    from animate3d_amd.train import FlatAdamW
    ops = TorchRefOps()
    g = torch.Generator().manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in ((7, 5), (1,), (130,))]
    before = [p.detach().clone() for p in ps]
    opt = FlatAdamW(ps, ops, lr=1e-2, max_grad_norm=0.0)
    assert all(torch.equal(p.detach(), b) for p, b in zip(ps, before))      # re-homing keeps the values ...
    assert all(o % 64 == 0 for o in opt.offsets)                             # ... in 256-byte aligned slots of one buffer
    for p in ps:
        p.grad = None                                                        # optimizer.zero_grad(set_to_none=True) of a caller
    opt.zero_grad()
    loss = sum((p ** 2).sum() for p in ps)
    loss.backward()
    assert all(p.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * o for p, o in zip(ps, opt.offsets))
    opt.step()
    sd = opt.state_dict()
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt2 = FlatAdamW(ps2, ops, lr=1e-2, max_grad_norm=0.0)
    opt2.load_state_dict(sd)
    for o, pp in ((opt, ps), (opt2, ps2)):
        o.zero_grad()
        sum((p ** 2).sum() for p in pp).backward()
        o.step()
    assert opt2.step_count == 2 and all(torch.equal(a, b) for a, b in zip(ps, ps2))


def test_flat_adamw_is_driven_by_torch_lr_schedulers():
    """train.py:431-440, 603: get_scheduler(...) returns a LambdaLR over the optimiser; warm-up must reach the fused step."""
    from animate3d_amd.train import FlatAdamW
    ops = TorchRefOps()
    p = torch.nn.Parameter(torch.ones(10))
    q = torch.nn.Parameter(torch.ones(10))
    opt = FlatAdamW([p], ops, lr=1e-2, weight_decay=0.0, max_grad_norm=0.0)
    ref = torch.optim.AdamW([q], lr=1e-2, weight_decay=0.0)
    warm = lambda step: min(1.0, (step + 1) / 4)
    s1 = torch.optim.lr_scheduler.LambdaLR(opt, warm)
    s2 = torch.optim.lr_scheduler.LambdaLR(ref, warm)
    for _ in range(6):
        for o, t in ((opt, p), (ref, q)):
            o.zero_grad()
            (t ** 3).sum().backward()
            o.step()
        s1.step(); s2.step()
        assert abs(s1.get_last_lr()[0] - s2.get_last_lr()[0]) < 1e-12 and abs(opt.lr - s2.get_last_lr()[0]) < 1e-12
    assert torch.allclose(p.detach(), q.detach(), rtol=0, atol=1e-6)


# ------------------------------------------------------------------ the reference's own loop: torch DDP + torch.optim.AdamW around the drop-in
def _ddp_worker(rank, world, port, q):
    import os
    import sys
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, Fr, hw = 2, 2, (8, 8)
        ocfg, ref, model = _pair(n, Fr, hw)
        model.enable_training()
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(params, lr=1e-3)                             # train.py:351-357
        ddp = DDP(model)                                                     # train.py:456-457
        ddp.train()
        inp = O.synthetic_inputs(ocfg, n, n, Fr, hw, seed=40 + rank, cfg_doubled=False)      # a different shard per rank
        target = torch.randn(n, 4, Fr - 1, *hw, generator=torch.Generator().manual_seed(50 + rank))
        with ddp.no_sync():                                                  # local gradient, no exchange
            F.mse_loss(ddp(**inp).sample[:, :, 1:].float(), target).backward()
        local = torch.cat([p.grad.reshape(-1) for p in params]).clone()
        opt.zero_grad()
        loss = F.mse_loss(ddp(**inp).sample[:, :, 1:].float(), target)       # train.py:572-577
        loss.backward()                                                      # DDP's bucketed all-reduce (mean) rides on the Function graph
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)             # :593
        got = torch.cat([p.grad.reshape(-1) for p in params]).clone()
        opt.step()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        mean = sum(gathered) / world
        mean = mean * min(1.0, 1.0 / (float(mean.norm()) + 1e-6))
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        peers = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(peers, flat)
        q.put((rank, float((got - mean).abs().max() / mean.abs().max()), all(torch.equal(peers[0], t) for t in peers)))
    except Exception as e:
        q.put((rank, repr(e), False))
        raise
    finally:
        dist.destroy_process_group()


def test_reference_training_loop_with_torch_ddp_over_gloo():
    """The unmodified pattern of train.py:351-357, 456-457, 572-596 — DistributedDataParallel around the UNet, torch.optim.AdamW,
    clip_grad_norm_ — works on the drop-in: DDP's gradient hooks fire on the parameters the autograd op set feeds."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, same in res:
        assert not isinstance(err, str), err
        print(f"[parity] rank {rank}: DDP-averaged, clipped gradient vs mean of the ranks' local gradients: {err:.2e}; parameters identical across ranks: {same}")
        assert err < 1e-3 and same          # two fp32 evaluations of the clip norm


def test_staged_training_refreezing_a_module_drops_its_stale_frozen_pack():
    """Staged training (train a module, then freeze it and train another in the same process): the persistent pack of the FROZEN
    sub-modules is valid for one set of trainable parameters only.  After the motion modules were trained and then frozen by
    ``requires_grad_(False)`` (no ``_invalidate()`` involved), the next grad-enabled forward must read their UPDATED weights — i.e. equal
    the ``no_grad`` forward, which always re-packs — and not the weights of the first pack."""
    ocfg, ref, model = _pair(2, 2, (8, 8))
    inp = O.synthetic_inputs(ocfg, 2, 2, 2, (8, 8), seed=3, cfg_doubled=False)
    model.enable_training()
    y0 = model(**inp).sample                              # builds the frozen pack (motion modules + i2v trainable)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "motion_modules." in name and p.ndim >= 2:
                p.mul_(1.5)                               # "training" moved the motion modules
    for name, p in model.named_parameters():              # stage 2: freeze them, keep i2v trainable
        if "motion_modules." in name:
            p.requires_grad_(False)
    y_train = model(**inp).sample
    with torch.no_grad():
        y_eval = model(**inp).sample
    assert not torch.allclose(y_train, y0)                # the new weights are seen ...
    # ... as the re-packing inference path sees them (that path runs the merged out-projections of round 5 — [t | sp | img] x
    # [ct Wo | cs Wsp | ci Wimg] in one fp32 sum — so the two agree to fp32 summation order, not bit for bit; stale weights are O(1) off)
    torch.testing.assert_close(y_train.detach(), y_eval, rtol=1e-3, atol=1e-4)


def test_trainable_merge_weights_are_read_back_once_per_forward(monkeypatch):
    """The kernels take the AlphaBlender weight by value, so a trainable ``mix_factor`` has to be read back from the device; one
    ``float(tensor)`` per merge GEMM would be a host synchronisation in the middle of the forward (~130 per training step at the real
    depth).  ``AutogradOps.prefetch_scalars`` reads all of them in one transfer while the step's weights are packed: no merge GEMM may fall
    back to its own read-back, and the values must be the ones the gradient check of the oracle test sees (same forward result)."""
    from animate3d_amd import autograd_ops as A
    n, Fr, hw = 2, 2, (8, 8)
    ocfg, ref, model = _pair(n, Fr, hw)
    inp = O.synthetic_inputs(ocfg, n, n, Fr, hw, seed=3, cfg_doubled=False)
    model.enable_training()
    aops = model._autograd_ops()
    fallbacks, batches = [], []
    real_prefetch = A.AutogradOps.prefetch_scalars

    def counting_prefetch(self, tensors):
        batches.append(len([t for t in tensors if torch.is_tensor(t)]))
        return real_prefetch(self, tensors)

    def counting_scalar(self, t):
        hit = self._host_scalars.get(id(t))
        if hit is None or hit[0] is not t:
            fallbacks.append(t)
        return float(t.detach())

    monkeypatch.setattr(A.AutogradOps, "prefetch_scalars", counting_prefetch)
    monkeypatch.setattr(A.AutogradOps, "_scalar", counting_scalar)
    y = model(**inp).sample
    assert batches and batches[0] > 0 and len(batches) == 1            # one batched read-back for the whole forward
    assert not fallbacks                                                # every tensor-valued merge weight was in it
    want = ref(**inp).sample
    assert float((y.detach() - want).abs().max()) < 2e-3 * float(want.abs().max())
    assert all(abs(v - float(t.detach())) == 0.0 for t, v in aops._host_scalars.values())


def test_gradient_checkpointing_recomputes_layers_and_changes_no_gradient():
    """train.py:381-382 ``unet.enable_gradient_checkpointing()``: one checkpoint per layer (ResNet + Transformer2D + motion module).  The
    recomputation runs the same ops through the same autograd functions, so loss and every parameter gradient must equal the
    un-checkpointed ones bit for bit, while the saved-tensor count of the graph drops (only layer boundaries are kept)."""
    n, Fr, hw = 2, 2, (8, 8)
    ocfg, ref, model = _pair(n, Fr, hw)
    inp = O.synthetic_inputs(ocfg, n, n, Fr, hw, seed=3, cfg_doubled=False)
    target = torch.randn(n, 4, Fr, *hw, generator=torch.Generator().manual_seed(1))
    from animate3d_amd.train import select_trainable
    select_trainable(model)
    model.enable_training()

    def run():
        for p in model.parameters():
            p.grad = None
        packed = []
        with torch.autograd.graph.saved_tensors_hooks(lambda t: (packed.append(t.numel()), t)[1], lambda t: t):
            loss = F.mse_loss(model(**inp).sample, target)
        loss.backward()
        return float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters() if p.requires_grad}, sum(packed)

    loss0, g0, kept0 = run()
    model.enable_gradient_checkpointing()
    loss1, g1, kept1 = run()
    model.disable_gradient_checkpointing()
    assert loss0 == loss1 and g0.keys() == g1.keys() and len(g0) > 50
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    print(f"[parity] elements saved for the backward: {kept0} without, {kept1} with gradient checkpointing")
    assert kept1 < 0.5 * kept0
    with torch.no_grad():                                   # the inference path is untouched
        model.enable_gradient_checkpointing()
        assert torch.isfinite(model(**inp).sample).all()


def test_packed_weight_with_a_sink_consumer_and_a_sliced_consumer_keeps_both_gradients():
    """PackW: the GEMM on the full packed operand leaves its fp32 weight gradient in the sink and returns a stride-0 placeholder; a GEMM on
    a SLICE of the same operand (unet._mv_attention's first-frame K|V projection under frame sharding: ``w_kvq[:2 * C]``) has no sink and
    returns a real gradient through autograd.  Both must reach the master parameters."""
    from animate3d_amd import autograd_ops as A
    torch.manual_seed(0)
    m1 = torch.randn(6, 8, requires_grad=True)
    m2 = torch.randn(4, 8, requires_grad=True)
    w = A.pack_weight(torch.float32, [m1, m2])
    x = torch.randn(5, 8)

    class SinkGemm(torch.autograd.Function):          # what autograd_ops' GEMM does for a packed operand
        @staticmethod
        def forward(ctx, x_, w_, sink):
            ctx.save_for_backward(x_, w_)
            ctx.sink = sink
            return x_ @ w_.t()

        @staticmethod
        def backward(ctx, dy):
            x_, w_ = ctx.saved_tensors
            return None, A._weight_grad(ctx.sink, w_, dy.t() @ x_), None

    y_full = SinkGemm.apply(x, w, w._a3d_sink)
    y_slice = x @ w[:6].t()                           # plain consumer of a view: no sink
    (y_full.sum() * 2.0 + y_slice.sum() * 3.0).backward()
    w_ref = torch.cat([m1.detach(), m2.detach()], 0).requires_grad_(True)
    ((x @ w_ref.t()).sum() * 2.0 + (x @ w_ref[:6].t()).sum() * 3.0).backward()
    assert torch.allclose(m1.grad, w_ref.grad[:6], atol=1e-6) and torch.allclose(m2.grad, w_ref.grad[6:], atol=1e-6)
    assert w._a3d_sink.grad32 is None


def test_weight_rows_slice_keeps_the_placeholder_memory_free(monkeypatch):
    """a ``weight_rows`` slice whose consumer is the FIRST sink consumer of the pass hands autograd the stride-0 zero placeholder;
    autograd's plain slice backward would turn it into a dense full-size zero tensor, and PackW.backward would add that to the sink's gradient
    (a full-width fp32 materialisation + add per pass).  ``weight_rows`` is its own autograd node now: what reaches PackW.backward is stride 0."""
    from animate3d_amd import autograd_ops as A
    torch.manual_seed(0)
    m1 = torch.randn(6, 8, requires_grad=True)
    m2 = torch.randn(4, 8, requires_grad=True)
    w = A.pack_weight(torch.float32, [m1, m2])
    x = torch.randn(5, 8)

    class SinkGemm(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x_, w_, sink):
            ctx.save_for_backward(x_, w_)
            ctx.sink = sink
            return x_ @ w_.t()

        @staticmethod
        def backward(ctx, dy):
            x_, w_ = ctx.saved_tensors
            return None, A._weight_grad(ctx.sink, w_, dy.t() @ x_), None

    seen = []
    real_backward = A.PackW.backward

    def spy(ctx, dw):
        seen.append(None if dw is None else tuple(dw.stride()))
        return real_backward(ctx, dw)
    monkeypatch.setattr(A.PackW, "backward", staticmethod(spy))
    rows = A.weight_rows(w, 0, 6)                      # the slice consumer runs its backward FIRST (it is the later op of the forward)
    y_full = SinkGemm.apply(x, w, w._a3d_sink)
    y_rows = SinkGemm.apply(x, rows, rows._a3d_sink)
    (y_full.sum() * 2.0 + y_rows.sum() * 3.0).backward()
    w_ref = torch.cat([m1.detach(), m2.detach()], 0).requires_grad_(True)
    ((x @ w_ref.t()).sum() * 2.0 + (x @ w_ref[:6].t()).sum() * 3.0).backward()
    assert torch.allclose(m1.grad, w_ref.grad[:6], atol=1e-6) and torch.allclose(m2.grad, w_ref.grad[6:], atol=1e-6)
    assert seen and all(st is None or all(v == 0 for v in st) for st in seen), seen      # never a dense tensor


def test_split_cols_collects_the_piece_gradients_in_one_buffer():
    """AutogradOps.split_cols on a fused K | V | Q | Q_i2v projection output: Q and Q_i2v feed one attention each (their dQ is written
    straight into the columns of the projection output's gradient buffer), K and V feed both (autograd sums the two dK / dV, the sum is
    copied in), an unused piece comes back zero.  Same gradient as plain slicing."""
    from animate3d_amd.autograd_ops import AutogradOps
    from animate3d_amd.hip_ops import RowMap
    torch.manual_seed(0)
    base = TorchRefOps(torch.float32, torch.device("cpu"))
    aops = AutogradOps(base)
    heads, C, groups, S = 2, 16, 3, 10
    rows = groups * S
    m = RowMap(gdiv=1, ga=S, gb=0, seg_len=S, seg_stride=0)
    x0 = torch.randn(rows, 5 * C)
    wo = torch.randn(rows, C)

    calls = []
    real_bwd = base.flash_attn_bwd

    def spy(*a, **k):
        calls.append({n: k.get(n) is not None for n in ("dq_out", "dk_out", "dv_out")})
        return real_bwd(*a, **k)
    base.flash_attn_bwd = spy

    def run(split):
        x = x0.clone().requires_grad_(True)
        h = x * 1.0                                   # a non-leaf, like the projection GEMM's output
        k, v, q, qi, unused = split(h)
        a = aops.flash_attn(q, k, v, m, m, groups, heads, S, S)
        ai = aops.flash_attn(qi, k, v, m, m, groups, heads, S, S)
        ((a * wo).sum() + 2.0 * (ai * wo).sum()).backward()
        return x.grad

    g_plain = run(lambda h: tuple(h[:, i * C:(i + 1) * C] for i in range(5)))
    assert all(not any(c.values()) for c in calls)
    calls.clear()
    g_split = run(lambda h: aops.split_cols(h, *range(0, 5 * C + 1, C)))
    assert torch.allclose(g_split, g_plain, atol=1e-6)
    assert torch.equal(g_split[:, 4 * C:], torch.zeros(rows, C))
    # both attentions wrote their dQ in place; K / V have two consumers, so neither wrote dK / dV in place
    assert calls == [{"dq_out": True, "dk_out": False, "dv_out": False}] * 2

    # one consumer per piece (the motion module's spatial branch): dQ, dK and dV are all written in place
    calls.clear()
    x = x0[:, :3 * C].clone().requires_grad_(True)
    k, v, q = aops.split_cols(x * 1.0, 0, C, 2 * C, 3 * C)
    (aops.flash_attn(q, k, v, m, m, groups, heads, S, S) * wo).sum().backward()
    assert calls == [{"dq_out": True, "dk_out": True, "dv_out": True}]
    xr = x0[:, :3 * C].clone().requires_grad_(True)
    (aops.flash_attn(xr[:, 2 * C:], xr[:, :C], xr[:, C:2 * C], m, m, groups, heads, S, S) * wo).sum().backward()
    assert torch.allclose(x.grad, xr.grad, atol=1e-6)
    # no gradient wanted: plain views, nothing recorded
    with torch.no_grad():
        assert not hasattr(aops.split_cols(x0, 0, C, 5 * C)[0], "_a3d_gcols")


def test_deferred_param_grads_equal_autograd_accumulation():
    """``autograd_ops.deferred_param_grads()`` (what train.training_step wraps its backward in): the weight / bias / affine gradients that
    PackW and the kernel backwards park are added to ``.grad`` in one multi-tensor call after the pass — the same numbers AccumulateGrad
    leaves, nothing parked outside the context, nothing left parked after it."""
    from animate3d_amd import autograd_ops as A
    ocfg, ref, model = _pair(2, 2, (8, 8), motion_image_attn=True)
    inp = O.synthetic_inputs(ocfg, 2, 2, 2, (8, 8), seed=3, cfg_doubled=False)
    target = torch.randn(2, 4, 1, 8, 8, generator=torch.Generator().manual_seed(1))
    model.enable_training()
    params = {k: p for k, p in model.named_parameters() if p.requires_grad}
    _loss(model, inp, target).backward()                       # plain: AccumulateGrad creates every .grad
    assert not A._Deferred.dst and not A._Deferred.active
    g0 = {k: p.grad.clone() for k, p in params.items()}
    assert len(g0) > 50 and sum(bool(v.abs().sum() > 0) for v in g0.values()) > len(g0) // 2
    for p in params.values():
        p.grad.zero_()

    parked = []
    real = torch._foreach_add_

    def spy(dst, src, *a, **k):
        parked.append(len(dst))
        return real(dst, src, *a, **k)
    torch._foreach_add_ = spy
    try:
        with A.deferred_param_grads():
            _loss(model, inp, target).backward()
            assert A._Deferred.active and len(A._Deferred.dst) > 50
    finally:
        torch._foreach_add_ = real
    assert parked and parked[0] > 50 and not A._Deferred.dst and not A._Deferred.src and not A._Deferred.active
    for k, p in params.items():
        assert torch.equal(p.grad, g0[k]), k

    # an exception inside the context drops what was parked instead of adding half a pass
    with pytest.raises(RuntimeError, match="stop"):
        with A.deferred_param_grads():
            gpar = next(iter(params.values())).grad
            A._Deferred.park(gpar, torch.ones_like(gpar))
            raise RuntimeError("stop")
    assert not A._Deferred.dst and not A._Deferred.src and not A._Deferred.active and A._Deferred.parked_bytes == 0

    # parked slices are flushed in chunks (bounded extra peak memory), with the same sums
    for p in params.values():
        p.grad.zero_()
    old, A._Deferred.FLUSH_BYTES = A._Deferred.FLUSH_BYTES, 4096
    parked.clear()
    torch._foreach_add_ = spy
    try:
        with A.deferred_param_grads():
            _loss(model, inp, target).backward()
    finally:
        torch._foreach_add_ = real
        A._Deferred.FLUSH_BYTES = old
    assert len(parked) > 3
    for k, p in params.items():
        assert torch.equal(p.grad, g0[k]), k


def test_rowmap_coverage_decides_the_zero_fill_of_dk_dv():
    """hip_ops.flash_attn_bwd skips the zero fill of dK / dV when the key map reaches every row of K / V exactly once: true for the
    multi-view map "(b n f) l -> (b f) (n l)" (attention_processor.py:340), false for the first-frame maps (frame 0's keys only, :389-418),
    for short key lengths and for maps that read a row twice."""
    from animate3d_amd.hip_ops import RowMap, rowmap_covers, rowmap_rows
    b, n, Fr, L = 2, 3, 4, 5
    rows = b * n * Fr * L
    mv = RowMap(gdiv=Fr, ga=n * Fr * L, gb=L, seg_len=L, seg_stride=Fr * L)            # unet._mv_maps: queries and keys of (b, f)
    k0 = RowMap(gdiv=Fr, ga=n * Fr * L, gb=0, seg_len=L, seg_stride=Fr * L)            # same, frame 0 of every b
    idx = rowmap_rows(mv, b * Fr, 1, n * L)
    want = torch.arange(rows).view(b, n, Fr, L).permute(0, 2, 1, 3).reshape(-1)         # "(b n f) l -> (b f) (n l)"
    assert torch.equal(idx, want)
    assert rowmap_covers(mv, b * Fr, 1, n * L, rows)
    assert not rowmap_covers(k0, b * Fr, 1, n * L, rows)                                # every frame's group reads frame 0: rows repeat
    assert not rowmap_covers(k0, b * Fr, Fr, n * L, rows)                               # one reader per video: frames > 0 never read
    assert rowmap_covers(k0, b * Fr, Fr, n * L, b * n * L) is False                     # (the rows it reads are not 0 .. b n L - 1 either)
    assert not rowmap_covers(mv, b * Fr, 1, n * L - 1, rows)                            # ragged: fewer keys than rows
    assert not rowmap_covers(mv, b * Fr, 1, n * L, rows + L)                            # a tensor with rows the map never reaches
    flat = RowMap(gdiv=1, ga=7, gb=0, seg_len=7, seg_stride=0)                          # plain [groups, 7] layout
    assert rowmap_covers(flat, 6, 1, 7, 42) and not rowmap_covers(flat, 6, 2, 7, 42)
    # the per-view first-frame map of the motion module's image branch (unet._motion_attn): V videos, frame 0 of each
    V = b * n
    img0 = RowMap(gdiv=Fr, ga=Fr * L, gb=0, seg_len=L, seg_stride=0)
    assert torch.equal(rowmap_rows(img0, V * Fr, Fr, L), (torch.arange(V).view(-1, 1) * Fr * L + torch.arange(L).view(1, -1)).reshape(-1))
    assert not rowmap_covers(img0, V * Fr, Fr, L, V * Fr * L)
