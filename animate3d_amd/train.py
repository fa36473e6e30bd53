"""Training step of the MV-VDM UNet on MI355X (reference train.py:343-357 optimiser set-up, :540-601 step).

What the reference does per step — sample noise / timesteps, keep the first frame clean, predict the noise with the UNet under
autocast, MSE on the noisy frames, ``scaler.scale(loss).backward()``, ``clip_grad_norm_``, ``AdamW.step`` with DDP's gradient
all-reduce in between — is kept call for call; what changes is where the work runs:

* forward and backward of the UNet: the HIP kernels behind ``MVUNetMotionModel.enable_training()`` (``autograd_ops.py``);
* optimiser: every trainable parameter is re-homed as a view of ONE flat fp32 buffer (and its ``.grad`` as a view of one flat
  gradient buffer), so gradient norm, clipping, loss-scale handling and the AdamW update are three kernel launches
  (``a3d_sqnorm_f32`` / ``a3d_clip_ctrl_f32`` / ``a3d_adamw_f32``) instead of ~2 500 small torch ops, and the data-parallel
  gradient exchange is a handful of large RCCL all-reduces over that same buffer (xGMI rings are per-link bound: few, large
  messages) instead of DDP's 25 MB buckets;
* one process per GPU, batch sharded over ranks (the reference's DistributedSampler), no sharding of the model.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .autograd_ops import deferred_param_grads
from .denoise import randn_like_reference

TRAINABLE_MODULES = ("i2v.", "motion_modules.")        # configs/training/train.yaml: trainable_modules


def select_trainable(unet, trainable_modules: Sequence[str] = TRAINABLE_MODULES):
    """train.py:343-351: freeze everything, then re-enable the parameters whose name contains one of ``trainable_modules``."""
    unet.requires_grad_(False)
    out = []
    for name, p in unet.named_parameters():
        if any(t in name for t in trainable_modules):
            p.requires_grad = True
            out.append(p)
    return out


class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (train.py:351-357) over one flat buffer, with clip_grad_norm_ (train.py:586, 593) and the
    GradScaler's dynamic loss scale (train.py:583-590; only needed for fp16 kernels) folded into the step.

    ``params`` keep being ordinary ``nn.Parameter`` objects (state_dict, checkpointing and the model code see no difference);
    their storage and their ``.grad`` are views of ``self.flat_p`` / ``self.flat_g``.  A ``torch.optim.Optimizer`` with one
    parameter group, so the learning-rate schedulers of train.py:431-440 (``diffusers.optimization.get_scheduler`` -> ``LambdaLR``)
    drive it through ``param_groups[0]["lr"]``; its ``state_dict`` holds the flat moment buffers."""

    def __init__(self, params: Iterable[torch.nn.Parameter], ops, lr: float = 1e-4, betas=(0.9, 0.999), weight_decay: float = 1e-2,
                 eps: float = 1e-8, max_grad_norm: float = 1.0, loss_scale: Optional[float] = None, growth_interval: int = 2000,
                 bucket_bytes: int = 512 << 20):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameter")
        dev = self.params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise ValueError("FlatAdamW keeps fp32 master parameters on one device (the reference trains an fp32 model under autocast)")
        super().__init__(self.params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.ops = ops
        self.max_grad_norm = max_grad_norm
        # every parameter starts on a 256-byte boundary of the flat buffers (the kernels read affine vectors with 16-byte loads; a
        # 1-element mix_factor would otherwise misalign everything behind it); the padding stays zero in all four buffers
        ALIGN = 64
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                p.grad = self.flat_g[off:off + k].view(p.shape)       # autograd accumulates into this view in place
        self.step_count = 0
        self.loss_scale = loss_scale              # None: no scaling (bf16 kernels)
        self.growth_interval, self._good_steps = growth_interval, 0
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.last_ctrl = None

    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    def zero_grad(self, set_to_none: bool = False):
        """Zeroes the flat gradient and keeps every ``.grad`` a view of it (``set_to_none`` is accepted and ignored: autograd must keep
        accumulating into the flat buffer)."""
        self.flat_g.zero_()
        for p, (off, k) in zip(self.params, self._spans()):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:      # somebody replaced .grad (set_to_none)
                p.grad = self.flat_g[off:off + k].view(p.shape)

    def _spans(self):
        for p, off in zip(self.params, self.offsets):
            yield off, p.numel()

    def _check_homes(self) -> None:
        """The fused step updates the flat buffers: every parameter (and its gradient) must still live there.  ``unet.to()`` / ``.half()``
        after construction re-binds ``p.data``; ``model.zero_grad(set_to_none=True)`` without ``optimizer.zero_grad()`` drops ``p.grad``."""
        es = self.flat_p.element_size()
        for p, off in zip(self.params, self.offsets):
            if p.data_ptr() != self.flat_p.data_ptr() + off * es:
                raise RuntimeError("FlatAdamW: a parameter no longer aliases the flat parameter buffer (the model was moved / cast after the "
                                   "optimiser was built: build the optimiser after unet.to(...), as train.py:456 orders it)")
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + off * es:
                raise RuntimeError("FlatAdamW: a gradient no longer aliases the flat gradient buffer (call optimizer.zero_grad(), not "
                                   "model.zero_grad(set_to_none=True), between steps)")

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        return loss if self.loss_scale is None else loss * self.loss_scale

    def all_reduce_grads(self, group=None) -> int:
        """Sum the flat gradient over the data-parallel group in a few large collectives (the 1 / world average is applied inside
        the step's gradient factor).  Returns the world size."""
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        world = dist.get_world_size(group)
        if world == 1:
            return 1
        works = [dist.all_reduce(self.flat_g[i:i + self.bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=True)
                 for i in range(0, self.flat_g.numel(), self.bucket_elems)]
        for w in works:
            w.wait()
        return world

    def step(self, world: int = 1) -> Dict[str, float]:
        """Unscale (+ average over ``world`` ranks), clip to ``max_grad_norm``, AdamW; skipped when the gradient is not finite.
        One host read-back per step (the three control floats), as GradScaler.update() has."""
        self._check_homes()
        self.step_count += 1
        inv = 1.0 / (world * (self.loss_scale if self.loss_scale is not None else 1.0))
        ctrl = self.ops.clip_ctrl(self.ops.sqnorm(self.flat_g), self.max_grad_norm, inv)
        grp = self.param_groups[0]
        self.ops.adamw_(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, lr=float(grp["lr"]), betas=tuple(grp["betas"]), eps=grp["eps"],
                        weight_decay=grp["weight_decay"], step=self.step_count, ctrl=ctrl)
        factor, skipped, norm = (float(x) for x in ctrl.tolist())
        self.last_ctrl = ctrl
        if skipped:
            self.step_count -= 1                     # torch.optim is not stepped when GradScaler finds an inf
            if self.loss_scale is not None:
                self.loss_scale *= 0.5
                self._good_steps = 0
        elif self.loss_scale is not None:
            self._good_steps += 1
            if self._good_steps >= self.growth_interval:
                self.loss_scale *= 2.0
                self._good_steps = 0
        return {"grad_norm": norm, "skipped": bool(skipped), "grad_factor": factor}

    def state_dict(self):
        return {"step": self.step_count, "lr": self.param_groups[0]["lr"], "initial_lr": self.param_groups[0].get("initial_lr"), "exp_avg": self.exp_avg[:self.numel].clone(), "exp_avg_sq": self.exp_avg_sq[:self.numel].clone(),
                "loss_scale": self.loss_scale, "good_steps": self._good_steps}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg[:self.numel].copy_(sd["exp_avg"])
        self.exp_avg_sq[:self.numel].copy_(sd["exp_avg_sq"])
        self.loss_scale, self._good_steps = sd["loss_scale"], int(sd["good_steps"])
        if sd.get("lr") is not None:
            self.param_groups[0]["lr"] = sd["lr"]
        if sd.get("initial_lr") is not None:
            self.param_groups[0]["initial_lr"] = sd["initial_lr"]


def add_noise(x: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor, alphas_cumprod: torch.Tensor) -> torch.Tensor:
    """DDIMScheduler.add_noise (train.py:553): sqrt(a_t) x + sqrt(1 - a_t) eps, a_t per batch element."""
    a = alphas_cumprod.to(x.device)[timesteps].to(x.dtype)
    shape = (-1,) + (1,) * (x.dim() - 1)
    return a.sqrt().reshape(shape) * x + (1.0 - a).sqrt().reshape(shape) * noise


def _sample_batch(latents: torch.Tensor, text_embeds: torch.Tensor, alphas_cumprod: torch.Tensor, generator, noise, timesteps):
    """train.py:540-569: noise on every frame but the first, one timestep per batch element, text states repeated per view."""
    b, n, c, f, h, w = latents.shape
    dev = latents.device
    first, rest = latents[:, :, :, 0:1], latents[:, :, :, 1:]                               # :540-543 first frame stays clean
    if noise is None:
        noise = randn_like_reference(rest.shape, generator, dev, rest.dtype)              # a CPU generator is the usual idiom
    if timesteps is None:
        gdev = generator.device if generator is not None else dev
        timesteps = torch.randint(0, alphas_cumprod.shape[0], (b,), generator=generator, device=gdev).long().to(dev)
    noisy = torch.cat([first, add_noise(rest, noise, timesteps, alphas_cumprod)], dim=3)      # :553-555
    noisy = noisy.reshape(b * n, c, f, h, w)
    ehs = text_embeds[:, None].expand(b, n, *text_embeds.shape[1:]).reshape(b * n, *text_embeds.shape[1:])      # :565-566
    t = timesteps[:, None].expand(b, n).reshape(b * n)                                          # :568-569
    return noisy, t, ehs, noise


def _loss(unet, noisy, t, ehs, cameras, image_embeds, noise, shape, num_views: int, i2v_cond_time_zero: bool) -> torch.Tensor:
    """train.py:572-577: UNet prediction, MSE against the noise on the noisy frames."""
    b, n, c, f, h, w = shape
    added = None if image_embeds is None else {"image_embeds": image_embeds}
    pred = unet(noisy, t, encoder_hidden_states=ehs, camera=cameras, num_views=num_views, added_cond_kwargs=added,
                i2v_cond_time_zero=i2v_cond_time_zero).sample                                   # :572-573
    pred = pred.reshape(b, n, c, f, h, w)[:, :, :, 1:]
    return F.mse_loss(pred.float(), noise.float(), reduction="mean")                          # :576-577


def training_step(unet, optimizer: FlatAdamW, latents: torch.Tensor, text_embeds: torch.Tensor, cameras: torch.Tensor,
                  image_embeds: Optional[torch.Tensor], *, alphas_cumprod: torch.Tensor, num_views: int, i2v_cond_time_zero: bool = False,
                  generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None,
                  timesteps: Optional[torch.Tensor] = None, group=None) -> Dict[str, float]:
    """One optimisation step, train.py:538-596.  ``latents`` [b, n, c, f, h, w] (scaled VAE latents), ``text_embeds`` [b, 77, 768],
    ``cameras`` [(b n), 16], ``image_embeds`` [(b n), 1024] or None.  ``unet.enable_training()`` must have been called."""
    noisy, t, ehs, noise = _sample_batch(latents, text_embeds, alphas_cumprod, generator, noise, timesteps)
    optimizer.zero_grad()
    loss = _loss(unet, noisy, t, ehs, cameras, image_embeds, noise, latents.shape, num_views, i2v_cond_time_zero)
    with deferred_param_grads():          # weight gradients land in the flat buffer in one multi-tensor add after the pass
        optimizer.scale(loss).backward()
    world = optimizer.all_reduce_grads(group)
    info = optimizer.step(world)
    info["loss"] = float(loss.detach())
    return info
