"""Single-node multi-GPU execution of one denoise step: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" for the CPU tests).

The reference has no inference parallelism at all; this is new design (§8e).  The (b, n, f) = (CFG half /
batch, view, frame) grid of videos-by-frames is cut along three axes, ``world = cfg_shards x view_shards x frame_shards``:

* axis 1 — CFG / batch halves ``b``: every regrouping in the model keeps ``b`` outermost
  (attention_processor.py:340,557), so different ``b`` never exchange data.  Free.
* axis 2 — views: a rank holds ``n / Sv`` views of its ``b`` slice.  The only exchange is an all-gather over the Sv ranks of
  the view group before each multi-view attention (16 Transformer2D + 42 motion-module spatial attentions per step) — of the
  attention's normalised input tokens (C wide; K|V are then projected locally for all n views), or optionally of the
  projected K|V (2C wide, ``gather_tokens = False``).  The I2V branch reuses the gathered K/V.
* axis 3 — frames: a rank holds a contiguous range of ``F / Sf`` frames.  Multi-view attention, convs, 2-D norms and
  cross-attention are local per (b, f); what crosses the frame group is
    - the temporal attention (attention_processor.py:619-641): all-gather of the projected K|V of every motion attention,
      queries stay local (``a3d_temporal_attn_sharded_bf16`` reads the rank-major gathered buffer in place);
    - the motion module's 3-D GroupNorm over (C/32, F, h, w) (diffusers TransformerTemporalModel.norm): fp64 partial sums
      [videos, 32, 2], one all-reduce (``a3d_group_norm_sums_bf16`` / ``a3d_group_norm_apply_bf16``);
    - the first-frame K/V of the I2V branches (attention_processor.py:389-397, 672-698): frame 0's normalised tokens are
      broadcast from the rank that holds frame 0 (1/F of a token tensor).

Rank order is (cfg, view, frame) with frame fastest: neighbouring ranks form a frame group.  On the fully connected xGMI mesh
every gather among S <= 8 peers uses all S-1 links of a GPU concurrently.

The model stays SPMD-transparent: every rank calls ``unet(...)`` with the full inputs and receives the full
``[V, C, F, h, w]`` output; the shard is cut inside ``forward`` and the (tiny) latent output is all-gathered at the end.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class ShardPlan:
    def __init__(self, group=None, layout: Optional[Sequence[int]] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if group is not None and group is not dist.group.WORLD:
            # sub-groups are created with dist.new_group (global ranks, collective over the WORLD): a plan over a foreign
            # group would build wrong rank lists or hang
            raise ValueError("ShardPlan works on the default process group (one process per GPU of the node)")
        self.group = None
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.layout_request = tuple(int(v) for v in layout) if layout is not None else None
        if self.layout_request is not None and (len(self.layout_request) != 3 or math.prod(self.layout_request) != self.world):
            raise ValueError(f"layout {self.layout_request} must be (cfg_shards, view_shards, frame_shards) with product {self.world}")
        self._layouts: Dict[Tuple[int, int, int], Tuple[object, object]] = {}
        self.cfg_shards = self.view_shards = self.frame_shards = 1
        self.cfg_rank = self.view_rank = self.frame_rank = 0
        self.view_group = self.frame_group = None
        self.frame_group_root = self.rank    # global rank holding frame 0 of this rank's frame group
        self.gather_bytes = 0            # bytes received by this rank in data-path collectives (telemetry)
        self.collectives = 0
        self.gather_tokens = True        # all-gather the attention's input tokens (C wide) instead of projected K|V (2C wide)
        # while a gather is in flight the persistent GEMM / conv kernels (one workgroup per CU) leave this many CUs to the
        # collective's own kernels: ``ops.reserved_cus`` of the HIP op set (set by shard_unet; None on CPU) is passed with every
        # a3d_gemm / a3d_conv3x3 launch as an explicit parameter — the library keeps no state, and a HIP graph captured while a
        # reservation is active simply replays with it
        self.reserve_cus = 16
        self.ops = None
        self._in_flight = 0

    # ---- layout: world = cfg_shards x view_shards x frame_shards
    @staticmethod
    def choose_layout(world: int, b: int, n: int, F: Optional[int] = None, request: Optional[Sequence[int]] = None) -> Tuple[int, int, int]:
        """Default: CFG halves first (free), then views, then frames.  ``request`` = explicit (cfg, views, frames)."""
        if request is not None:
            cfg, views, frames = (int(v) for v in request)
        else:
            cfg = math.gcd(world, b)
            views = math.gcd(world // cfg, n)
            frames = world // (cfg * views)
        if cfg * views * frames != world or b % cfg or n % views or (frames > 1 and (F is None or F % frames)):
            raise ValueError(f"cannot shard b={b} x n={n} views x F={F} frames over {world} ranks as "
                             f"(cfg_shards={cfg}, view_shards={views}, frame_shards={frames})")
        return cfg, views, frames

    def _global_rank(self, c: int, s: int, r: int) -> int:
        return (c * self.view_shards + s) * self.frame_shards + r

    def configure(self, b: int, n: int, F: Optional[int] = None):
        """Collective on first use of a layout: every rank must call it with the same (b, n, F).  Creates / reuses the view
        and frame groups (``dist.new_group`` is collective over the world — call it explicitly, e.g. through
        ``shard_unet(unet, shape=(b, n, F))``, if the first forward must not carry that hidden collective)."""
        cfg, views, frames = self.choose_layout(self.world, b, n, F, self.layout_request)
        self.release_reservation()           # a forward that died between *_start and *_finish must not leave CUs reserved
        self.cfg_shards, self.view_shards, self.frame_shards = cfg, views, frames
        r = self.rank
        self.frame_rank, self.view_rank, self.cfg_rank = r % frames, (r // frames) % views, r // (frames * views)
        key = (cfg, views, frames)
        if key not in self._layouts:
            vmine = fmine = None
            for c in range(cfg):
                for fr in range(frames):
                    ranks = [self._global_rank(c, s, fr) for s in range(views)]
                    g = dist.new_group(ranks=ranks) if views > 1 else None
                    if self.rank in ranks:
                        vmine = g
            for c in range(cfg):
                for s in range(views):
                    ranks = [self._global_rank(c, s, fr) for fr in range(frames)]
                    g = dist.new_group(ranks=ranks) if frames > 1 else None
                    if self.rank in ranks:
                        fmine = g
            self._layouts[key] = (vmine, fmine)
        self.view_group, self.frame_group = self._layouts[key]
        self.frame_group_root = self._global_rank(self.cfg_rank, self.view_rank, 0)
        return self

    def local_videos_on(self, V: int, n: int, device) -> torch.Tensor:
        """``local_videos`` as a tensor on ``device``, cached per (V, n, layout, device): the forward asks for it on every call, and a fresh
        host-to-device copy per call is a synchronising transfer (and illegal under HIP-graph capture)."""
        key = (V, n, self.cfg_shards, self.view_shards, self.cfg_rank, self.view_rank, str(device))
        cache = self.__dict__.setdefault("_idx_cache", {})
        if key not in cache:
            cache[key] = self.local_videos(V, n).to(device)
        return cache[key]

    def local_videos(self, V: int, n: int) -> torch.Tensor:
        """Indices (into the (b n) ordered video axis) of this rank's videos, in local (b n) order."""
        b = V // n
        bl, nl = b // self.cfg_shards, n // self.view_shards
        bs = torch.arange(self.cfg_rank * bl, (self.cfg_rank + 1) * bl)
        ns = torch.arange(self.view_rank * nl, (self.view_rank + 1) * nl)
        return (bs[:, None] * n + ns[None, :]).reshape(-1)

    def frame_range(self, F: int) -> Tuple[int, int]:
        """(first frame, number of frames) of this rank."""
        fl = F // self.frame_shards
        return self.frame_rank * fl, fl

    def _count(self, received_bytes: int, n: int = 1):
        self.gather_bytes += int(received_bytes)
        self.collectives += int(n)

    def _overlap(self, delta: int):
        """Book-keeping of asynchronous collectives in flight: the first one reserves CUs for RCCL, the last one to finish frees them."""
        before, self._in_flight = self._in_flight, max(0, self._in_flight + delta)
        if self.ops is not None and self.reserve_cus > 0 and (before == 0) != (self._in_flight == 0):
            self.ops.reserved_cus = self.reserve_cus if self._in_flight else 0

    def release_reservation(self):
        """Drop any CU reservation and forget collectives in flight (start of every configure(); error paths)."""
        self._in_flight = 0
        if self.ops is not None:
            self.ops.reserved_cus = 0

    # ---- view axis
    def all_gather_views_start(self, kv: torch.Tensor, b_local: int = 1):
        """Launch the all-gather of this rank's tokens (or projected K|V) ``[(b_l n_l f) l, width]`` over the view group and
        return a handle; the collective runs on the backend's own stream (RCCL), so kernels issued on the compute stream
        before ``all_gather_views_finish`` overlap with it (the Q projection, the temporal branch of a motion module)."""
        S = self.view_shards
        rows, width = kv.shape
        kv = kv.contiguous()
        out = torch.empty((S * rows, width), dtype=kv.dtype, device=kv.device)
        # one collective per local batch entry, each straight into its final place: rank s's rows of batch entry j land at
        # out[j][s] of the [b_l, S, per_b, width] view = unsharded (b_l N f) l order, so no re-ordering copy follows the gather
        # (b_local = 1: the single gather already is in that order)
        if b_local <= 0 or rows % b_local != 0:
            raise ValueError(f"all_gather_views: {rows} rows do not split into {b_local} local batch entries")
        per_b = rows // b_local
        works = []
        try:
            for j in range(b_local):
                works.append(dist.all_gather_into_tensor(out[j * S * per_b:(j + 1) * S * per_b], kv[j * per_b:(j + 1) * per_b],
                                                         group=self.view_group, async_op=True))
        except Exception:
            for w_ in works:
                w_.wait()
            raise
        self._count((S - 1) * rows * width * kv.element_size(), n=len(works))
        self._overlap(+1)
        return works, out, kv

    def all_gather_views_finish(self, handle, b_local: int) -> torch.Tensor:
        """Wait (the compute stream waits, not the host) and return the view group's ``[(b_l N f) l, width]`` tokens in
        unsharded row order."""
        works, out, kv = handle
        try:
            for w_ in works:
                w_.wait()
        finally:
            self._overlap(-1)
        return out

    def all_gather_views(self, kv: torch.Tensor, b_local: int) -> torch.Tensor:
        return self.all_gather_views_finish(self.all_gather_views_start(kv, b_local), b_local)

    # ---- frame axis
    def all_gather_frames_start(self, kv: torch.Tensor):
        """All-gather of the temporal K|V ``[(v f_l) l, 2C]`` over the frame group; the result is rank-major
        ``[Sf, (v f_l) l, 2C]`` and is read in place (frame f of video v sits in block f // f_l)."""
        S = self.frame_shards
        rows, width = kv.shape
        kv = kv.contiguous()
        out = torch.empty((S * rows, width), dtype=kv.dtype, device=kv.device)
        work = dist.all_gather_into_tensor(out, kv, group=self.frame_group, async_op=True)
        self._count((S - 1) * rows * width * kv.element_size())
        self._overlap(+1)
        return work, out, kv

    def all_gather_frames_finish(self, handle) -> torch.Tensor:
        work, out, _ = handle
        try:
            work.wait()
        finally:
            self._overlap(-1)
        return out

    def broadcast_frame0(self, x0: Optional[torch.Tensor], shape, dtype, device) -> torch.Tensor:
        """Frame 0's tokens from the rank of the frame group that holds frame 0 (``x0`` there, None elsewhere)."""
        buf = x0.contiguous() if self.frame_rank == 0 else torch.empty(shape, dtype=dtype, device=device)
        dist.broadcast(buf, src=self.frame_group_root, group=self.frame_group)
        if self.frame_rank != 0:
            self._count(buf.numel() * buf.element_size())
        return buf

    def all_reduce_frames(self, t: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.frame_group)
        self._count(2 * (self.frame_shards - 1) * t.numel() * t.element_size() // self.frame_shards)
        return t

    # ---- output
    def all_gather_output(self, y_local: torch.Tensor, V: int, n: int, F: Optional[int] = None) -> torch.Tensor:
        """[V_local, C, F_local, h, w] on every rank -> full [V, C, F, h, w] in (b n) order on every rank."""
        b = V // n
        bl, nl = b // self.cfg_shards, n // self.view_shards
        out = torch.empty((self.world * y_local.shape[0],) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
        dist.all_gather_into_tensor(out, y_local.contiguous())
        C, Fl, h, w = y_local.shape[1:]
        out = out.view(self.cfg_shards, self.view_shards, self.frame_shards, bl, nl, C, Fl, h, w)
        out = out.permute(0, 3, 1, 4, 5, 2, 6, 7, 8)                     # cfg, b_l, view, n_l, C, frame shard, F_l, h, w
        return out.reshape(V, C, self.frame_shards * Fl, h, w).contiguous()


ViewParallel = ShardPlan          # name of the round-1 plan (views only)


class RankShapePlan(ShardPlan):
    """The COMPUTE leg of rank 0 of layout (cfg, views, frames) in a single process, without a process group: the model cuts rank 0's
    shard out of the full call and launches exactly the kernels that rank launches — local rows, the K|V projection over the gathered
    token count, attention with q_len != kv_len through the unsharded row maps, the CU reservation while a "gather" is in
    flight — with every data-path collective replaced by local device copies of the right size (the peers' blocks are copies of this
    rank's block, so the outputs are NOT the sharded job's outputs; only the launch shapes and the timing are).  ``bench.py --rank-shape
    c,v,f`` measures a rank's compute time on the one GPU a builder has; link time is not in it."""

    def __init__(self, layout: Sequence[int]):      # no super().__init__: that one needs torch.distributed
        c, v, f = (int(s) for s in layout)
        if min(c, v, f) < 1:
            raise ValueError(f"rank shape {tuple(layout)} must be three positive shard counts (cfg, views, frames)")
        self.group = None
        self.rank = 0
        self.world = c * v * f
        self.layout_request = (c, v, f)
        self._layouts = {}
        self.cfg_shards, self.view_shards, self.frame_shards = c, v, f
        self.cfg_rank = self.view_rank = self.frame_rank = 0
        self.view_group = self.frame_group = None
        self.frame_group_root = 0
        self.gather_bytes = self.collectives = 0
        self.gather_tokens = True
        self.reserve_cus = 16
        self.ops = None
        self._in_flight = 0

    def configure(self, b: int, n: int, F: Optional[int] = None):
        self.choose_layout(self.world, b, n, F, self.layout_request)       # same divisibility errors as the real plan
        self.release_reservation()
        return self

    @staticmethod
    def _replicate(t: torch.Tensor, S: int, blocks: int = 1) -> torch.Tensor:
        """[blocks, S, rows / blocks, width] filled with S copies of every block: the gathered buffer's size and layout."""
        rows, width = t.shape
        out = torch.empty((S * rows, width), dtype=t.dtype, device=t.device)
        out.view(blocks, S, rows // blocks, width)[:] = t.view(blocks, 1, rows // blocks, width)
        return out

    def all_gather_views_start(self, kv: torch.Tensor, b_local: int = 1):
        if b_local <= 0 or kv.shape[0] % b_local != 0:
            raise ValueError(f"all_gather_views: {kv.shape[0]} rows do not split into {b_local} local batch entries")
        kv = kv.contiguous()
        self._count((self.view_shards - 1) * kv.numel() * kv.element_size(), n=b_local)
        self._overlap(+1)                # the reservation lasts until *_finish, as with the asynchronous collective
        return [], self._replicate(kv, self.view_shards, b_local), kv

    def all_gather_frames_start(self, kv: torch.Tensor):
        kv = kv.contiguous()
        self._count((self.frame_shards - 1) * kv.numel() * kv.element_size())
        self._overlap(+1)
        return _NoWork(), self._replicate(kv, self.frame_shards), kv

    def broadcast_frame0(self, x0, shape, dtype, device) -> torch.Tensor:
        return x0.contiguous()          # rank 0 holds frame 0: it is the sender

    def all_reduce_frames(self, t: torch.Tensor) -> torch.Tensor:
        self._count(2 * (self.frame_shards - 1) * t.numel() * t.element_size() // self.frame_shards)
        return t.mul_(self.frame_shards)          # sums over all frames ~ frame_shards x the local sums: keeps the statistics sane

    def all_gather_output(self, y_local: torch.Tensor, V: int, n: int, F: Optional[int] = None) -> torch.Tensor:
        C, Fl, h, w = y_local.shape[1:]
        reps = V // y_local.shape[0]
        return y_local.repeat(reps, 1, self.frame_shards, 1, 1).contiguous()


class _NoWork:
    def wait(self):
        return True


def rank_shape_unet(unet, layout: Sequence[int]) -> RankShapePlan:
    """Attach a RankShapePlan (single process, no torch.distributed): see the class.  Same op-set settings as ``shard_unet``."""
    unet.parallel = RankShapePlan(layout)
    ops = getattr(unet, "ops", None)
    unet.parallel.ops = ops if hasattr(ops, "reserved_cus") else None
    unet.parallel._restore_split_k = None
    return unet.parallel


def shard_unet(unet, group=None, layout: Optional[Sequence[int]] = None, shape: Optional[Tuple[int, int, int]] = None,
               bit_exact: bool = False) -> ShardPlan:
    """Attach a ShardPlan to a MVUNetMotionModel (every rank, same order).  ``layout`` = (cfg_shards, view_shards,
    frame_shards) or None for the default (CFG halves, then views, then frames); ``shape`` = (b, n, F) of the calls to come
    creates the process groups now instead of inside the first forward.  ``bit_exact``: sharded ranks see other row counts than the
    unsharded job, and split-K (HipOps.split_k: the 8 x 8 / 4 x 4 convolutions and small-M linears of levels 2 / 3) re-associates the K
    sum per launch shape — True turns it off for this op set until ``unshard_unet`` so that a CFG- / view-sharded forward equals the
    unsharded one (run with split_k off as well) BIT FOR BIT; the default keeps it: a rank of 8 runs its level-3 convolutions 4-10 x
    faster with it (profiles/README.md, round 6) and agrees with the unsharded job to fp32 summation order."""
    unet.parallel = ShardPlan(group, layout)
    ops = getattr(unet, "ops", None)                                # HIP op set: reserve CUs for RCCL while a gather overlaps the GEMMs
    unet.parallel.ops = ops if hasattr(ops, "reserved_cus") else None
    unet.parallel._restore_split_k = None
    if bit_exact and unet.parallel.ops is not None and hasattr(ops, "split_k"):
        unet.parallel._restore_split_k = bool(ops.split_k)
        ops.split_k = False
    if shape is not None:
        unet.parallel.configure(*shape)
    return unet.parallel


def unshard_unet(unet) -> None:
    """Detach the plan: drops any CU reservation and gives the op set its split-K setting back (the op set may be shared with other models)."""
    par = getattr(unet, "parallel", None)
    if par is None:
        return
    par.release_reservation()
    if getattr(par, "_restore_split_k", None) is not None and par.ops is not None:
        par.ops.split_k = par._restore_split_k
    unet.parallel = None
