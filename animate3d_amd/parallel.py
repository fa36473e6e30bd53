"""Single-node multi-GPU execution of one denoise step: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" for the CPU tests).

The reference has no inference parallelism at all; this is new design (§8e):

* axis 1 — CFG / batch halves ``b``: every regrouping in the model keeps ``b`` outermost
  (attention_processor.py:340,557), so different ``b`` never exchange data.  Free.
* axis 2 — views: rank (c, s) holds ``n/S`` of the ``n`` views of its ``b`` slice, all frames.
  Temporal attention, the 3-D GroupNorm, convs, GEMMs and cross-attention are local per video; the
  only exchange is an all-gather over the S ranks of the view group before each multi-view attention — of the
  attention's normalised input tokens (C wide; K|V are then projected locally for all n views), or optionally of
  the projected K|V tokens (2C wide, ``gather_tokens = False``) — (16 Transformer2D + 42 motion-module attentions per step; the
  I2V branch reuses the same gathered K/V).  On the fully connected xGMI mesh the gather among
  S <= 4 peers uses all S-1 links of a GPU concurrently.

The model stays SPMD-transparent: every rank calls ``unet(...)`` with the full inputs and receives
the full ``[V, C, F, h, w]`` output; the shard is cut inside ``forward`` and the (tiny) latent output
is all-gathered at the end.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.distributed as dist


class ViewParallel:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._layouts: Dict[Tuple[int, int], Tuple[list, object]] = {}
        self.cfg_shards = self.view_shards = 1
        self.cfg_rank = self.view_rank = 0
        self.view_group = None
        self.gather_bytes = 0            # bytes received by this rank in all-gathers (telemetry)
        self.gather_tokens = True        # all-gather the attention's input tokens (C wide) instead of projected K|V (2C wide)

    # ---- layout: world = cfg_shards x view_shards
    @staticmethod
    def choose_layout(world: int, b: int, n: int) -> Tuple[int, int]:
        cfg = math.gcd(world, b)
        views = world // cfg
        if n % views != 0:
            raise ValueError(f"cannot shard b={b} x n={n} views over {world} ranks (cfg_shards={cfg}, view_shards={views})")
        return cfg, views

    def configure(self, b: int, n: int):
        """Collective: every rank must call it with the same (b, n).  Creates / reuses the view groups."""
        cfg, views = self.choose_layout(self.world, b, n)
        key = (cfg, views)
        if key not in self._layouts:
            groups = []
            mine = None
            for c in range(cfg):
                ranks = [c * views + s for s in range(views)]
                g = dist.new_group(ranks=ranks) if views > 1 else None     # new_group is collective over the world
                groups.append(g)
                if self.rank in ranks:
                    mine = g
            self._layouts[key] = (groups, mine)
        self.cfg_shards, self.view_shards = cfg, views
        self.cfg_rank, self.view_rank = self.rank // views, self.rank % views
        self.view_group = self._layouts[key][1]
        return self

    def local_videos(self, V: int, n: int) -> torch.Tensor:
        """Indices (into the (b n) ordered video axis) of this rank's videos, in local (b n) order."""
        b = V // n
        bl, nl = b // self.cfg_shards, n // self.view_shards
        bs = torch.arange(self.cfg_rank * bl, (self.cfg_rank + 1) * bl)
        ns = torch.arange(self.view_rank * nl, (self.view_rank + 1) * nl)
        return (bs[:, None] * n + ns[None, :]).reshape(-1)

    # ---- collectives
    def all_gather_views_start(self, kv: torch.Tensor):
        """Launch the all-gather of this rank's projected K|V tokens ``[(b_l n_l f) l, 2C]`` over the view group and
        return a handle; the collective runs on the backend's own stream (RCCL), so kernels issued on the compute
        stream before ``all_gather_views_finish`` overlap with it (the Q projection, the temporal branch of a motion
        module)."""
        S = self.view_shards
        rows, width = kv.shape
        kv = kv.contiguous()
        out = torch.empty((S * rows, width), dtype=kv.dtype, device=kv.device)
        work = dist.all_gather_into_tensor(out, kv, group=self.view_group, async_op=True)
        self.gather_bytes += (S - 1) * rows * width * kv.element_size()
        return work, out, kv

    def all_gather_views_finish(self, handle, b_local: int) -> torch.Tensor:
        """Wait (the compute stream waits, not the host) and return the view group's ``[(b_l N f) l, 2C]`` tokens in
        unsharded row order."""
        work, out, kv = handle
        work.wait()
        S = self.view_shards
        rows, width = kv.shape
        if b_local == 1:
            return out                               # [S, n_l F L, 2C] is already (N f) l order
        per_b = rows // b_local
        return out.view(S, b_local, per_b, width).permute(1, 0, 2, 3).reshape(S * rows, width)

    def all_gather_views(self, kv: torch.Tensor, b_local: int) -> torch.Tensor:
        return self.all_gather_views_finish(self.all_gather_views_start(kv), b_local)

    def all_gather_output(self, y_local: torch.Tensor, V: int, n: int) -> torch.Tensor:
        """[V_local, C, F, h, w] on every rank -> full [V, C, F, h, w] in (b n) order on every rank."""
        b = V // n
        bl, nl = b // self.cfg_shards, n // self.view_shards
        out = torch.empty((self.world * y_local.shape[0],) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
        dist.all_gather_into_tensor(out, y_local.contiguous(), group=self.group)
        tail = tuple(y_local.shape[1:])
        out = out.view(self.cfg_shards, self.view_shards, bl, nl, *tail).permute(0, 2, 1, 3, *range(4, 4 + len(tail)))
        return out.reshape(V, *tail).contiguous()


def shard_unet(unet, group=None) -> ViewParallel:
    """Attach a ViewParallel plan to a MVUNetMotionModel (every rank, same order)."""
    unet.parallel = ViewParallel(group)
    return unet.parallel
