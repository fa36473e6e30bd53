"""world_size-2/4/8 gloo tests (CPU) of the multi-GPU path: the CFG / view / frame-sharded forward must equal the
single-process forward.  The host logic runs on the plain-torch op set of tests/torch_ops.py; what is
under test is the shard plan, the K|V all-gather layout and the output re-assembly of
animate3d_amd/parallel.py + the sharded branch of MVUNetMotionModel._mv_attention."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = dict(block_out_channels=(32, 64, 64, 64))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, F, hw, videos, layout, kw, cond0, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from animate3d_amd.config import UNetConfig
        from animate3d_amd.parallel import shard_unet
        from animate3d_amd.unet import MVUNetMotionModel
        from oracle import unet_ref as O
        from tests.torch_ops import TorchRefOps
        ocfg = O.UNetConfig(**SMALL, **kw)
        ref = O.MVUNetMotionModelRef(ocfg, n, F, hw).eval()
        O.init_synthetic_weights(ref, seed=0)
        model = MVUNetMotionModel(UNetConfig(**SMALL, **kw), ops=TorchRefOps(), num_views=n)
        model.load_state_dict(ref.state_dict())
        inp = O.synthetic_inputs(ocfg, videos, n, F, hw, seed=11, cfg_doubled=videos >= 2 * n)
        full = model(**inp, i2v_cond_time_zero=cond0).sample                     # unsharded, same process
        par = shard_unet(model, layout=layout, shape=(videos // n, n, F))
        sharded = model(**inp, i2v_cond_time_zero=cond0).sample                  # default: gathers the attention input tokens (C wide)
        tok_bytes = par.gather_bytes
        par.gather_tokens = False                      # alternative: gathers the projected K|V (2C wide)
        par.gather_bytes = 0
        sharded_kv = model(**inp, i2v_cond_time_zero=cond0).sample
        err = max((sharded - full).abs().max().item(), (sharded_kv - full).abs().max().item())
        if par.view_shards > 1:
            assert tok_bytes < par.gather_bytes
        q.put((rank, err, (par.cfg_shards, par.view_shards, par.frame_shards), par.gather_bytes, tuple(sharded.shape)))
    except Exception as e:   # surface the failure instead of letting the parent time out
        q.put((rank, repr(e), (0, 0, 0), 0, ()))
        raise
    finally:
        dist.destroy_process_group()


CASES = [   # world, n, F, videos, requested layout, expected (cfg, views, frames), config switches, i2v_cond_time_zero
    (2, 2, 2, 4, None, (2, 1, 1), {}, False),
    (2, 2, 2, 2, None, (1, 2, 1), {}, False),
    (4, 2, 2, 4, None, (2, 2, 1), {}, False),
    (8, 4, 2, 8, None, (2, 4, 1), {}, False),                     # bench.py --gpus 8 default: CFG x 4 view shards
    (4, 4, 2, 4, (1, 4, 1), (1, 4, 1), {}, False),                # BASELINE config 3's wording: 4 views on 4 GPUs
    (2, 2, 4, 2, (1, 1, 2), (1, 1, 2), {}, True),                 # frames only; frame 0 carries the t = 0 embedding
    (4, 2, 4, 4, (2, 1, 2), (2, 1, 2), {}, False),                # CFG x frames
    (4, 2, 4, 2, (1, 2, 2), (1, 2, 2), {}, False),                # views x frames (BASELINE config 4's wording)
    (8, 2, 4, 4, (2, 2, 2), (2, 2, 2), dict(motion_image_attn=True), True),   # all three axes + the first-frame image branch
    (2, 1, 4, 1, None, (1, 1, 2), dict(motion_spatial_attn=False), False),    # single view: the default falls through to frames
]


@pytest.mark.parametrize("world,n,F,videos,layout,expect,kw,cond0", CASES)
def test_sharded_forward_equals_unsharded(world, n, F, videos, layout, expect, kw, cond0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, F, (8, 8), videos, layout, kw, cond0, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, got_layout, gbytes, shape in res:
        assert not isinstance(err, str), err
        assert got_layout == expect
        assert shape == (videos, 4, F, 8, 8)
        assert err < 5e-4, (rank, err)   # fp32 summation-order noise through ~600 ops
        assert (gbytes > 0) == (expect[1] > 1 or expect[2] > 1)


def test_layout_choice_and_local_videos():
    from animate3d_amd.parallel import ShardPlan
    assert ShardPlan.choose_layout(8, 2, 4, 16) == (2, 4, 1)      # BASELINE config 2 on 8 GPUs: CFG x views
    assert ShardPlan.choose_layout(4, 2, 4, 16) == (2, 2, 1)
    assert ShardPlan.choose_layout(2, 2, 4, 16) == (2, 1, 1)
    assert ShardPlan.choose_layout(8, 2, 8, 32) == (2, 4, 1)      # config 4: 8 views
    assert ShardPlan.choose_layout(8, 2, 8, 32, (2, 2, 2)) == (2, 2, 2)
    assert ShardPlan.choose_layout(8, 1, 4, 16) == (1, 4, 2)      # no CFG axis: views, then frames
    assert ShardPlan.choose_layout(4, 2, 4, 16, (1, 4, 1)) == (1, 4, 1)
    with pytest.raises(ValueError):
        ShardPlan.choose_layout(8, 1, 4, 15)                      # frames do not divide
    with pytest.raises(ValueError):
        ShardPlan.choose_layout(8, 2, 4, 16, (2, 2, 1))           # product != world


# ---- bench.py --rank-shape: the single-process emulation of one rank must launch exactly what rank 0 of the real job launches
class _LoggedOps:
    """Delegates to TorchRefOps and records (op, tensor shapes, integer arguments) of every call."""

    def __init__(self, ops):
        object.__setattr__(self, "_ops", ops)
        object.__setattr__(self, "log", [])
        object.__setattr__(self, "reserved_log", [])

    def __setattr__(self, name, value):
        if name == "reserved_cus":
            self.reserved_log.append(int(value))
        setattr(self._ops, name, value)

    def __getattr__(self, name):
        fn = getattr(self._ops, name)
        if not callable(fn) or name.startswith("_") or name in ("empty", "split_cols", "interleave_geglu"):
            return fn

        def logged(*a, **k):
            sig = tuple(tuple(v.shape) if torch.is_tensor(v) else (v if isinstance(v, (int, bool, str)) else type(v).__name__) for v in a)
            ksig = tuple(sorted((kk, tuple(v.shape) if torch.is_tensor(v) else (v if isinstance(v, (int, bool)) else type(v).__name__)) for kk, v in k.items()))
            self.log.append((name, sig, ksig))
            return fn(*a, **k)
        return logged


def _launch_log_worker(rank, world, port, n, F, hw, videos, layout, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from animate3d_amd.config import UNetConfig
        from animate3d_amd.parallel import shard_unet
        from animate3d_amd.unet import MVUNetMotionModel
        from oracle import unet_ref as O
        from tests.torch_ops import TorchRefOps
        ops = _LoggedOps(TorchRefOps())
        ops.reserved_cus = 0
        model = MVUNetMotionModel(UNetConfig(**SMALL), ops=ops, num_views=n)
        model.init_synthetic(seed=0)
        inp = O.synthetic_inputs(O.UNetConfig(**SMALL), videos, n, F, hw, seed=11, cfg_doubled=True)
        shard_unet(model, layout=layout, shape=(videos // n, n, F))
        model(**inp)
        del ops.log[:]
        par = model.parallel
        par.gather_bytes = par.collectives = 0
        model(**inp)
        q.put((rank, list(ops.log) if rank == 0 else None, par.gather_bytes, par.collectives))
    except Exception as e:
        q.put((rank, repr(e), 0, 0))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,F,videos,layout", [(4, 2, 2, 4, (2, 2, 1)), (4, 2, 4, 2, (1, 2, 2)), (8, 2, 4, 4, (2, 2, 2))])
def test_rank_shape_emulation_launches_what_rank0_launches(world, n, F, videos, layout):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_launch_log_worker, args=(r, world, port, n, F, (8, 8), videos, layout, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    real = next(r for r in res if r[0] == 0)
    assert not isinstance(real[1], str), real[1]

    from animate3d_amd.config import UNetConfig
    from animate3d_amd.parallel import rank_shape_unet
    from animate3d_amd.unet import MVUNetMotionModel
    from oracle import unet_ref as O
    from tests.torch_ops import TorchRefOps
    ops = _LoggedOps(TorchRefOps())
    ops.reserved_cus = 0
    model = MVUNetMotionModel(UNetConfig(**SMALL), ops=ops, num_views=n)
    model.init_synthetic(seed=0)
    inp = O.synthetic_inputs(O.UNetConfig(**SMALL), videos, n, F, (8, 8), seed=11, cfg_doubled=True)
    plan = rank_shape_unet(model, layout)
    model(**inp)                                               # first call packs the weights (folding GEMMs), as in the worker
    del ops.log[:]
    plan.gather_bytes = plan.collectives = 0
    y = model(**inp).sample
    assert tuple(y.shape) == (videos, 4, F, 8, 8) and torch.isfinite(y).all()
    assert ops.log == real[1]                                  # same ops, same order, same shapes and integer arguments
    assert (plan.gather_bytes, plan.collectives) == (real[2], real[3])      # and the bytes / collectives rank 0 counts
    assert set(ops.reserved_log) == {0, plan.reserve_cus} and ops.reserved_log[-1] == 0      # reservation toggled around every "gather"
    with pytest.raises(ValueError):
        rank_shape_unet(model, (1, 3, 1)).configure(videos // n, n, F)      # the real plan's divisibility errors
