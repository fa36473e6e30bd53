"""world_size-2/4/8 gloo tests (CPU) of the multi-GPU path: the view/CFG-sharded forward must equal the
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


def _worker(rank, world, port, n, F, hw, videos, q):
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
        ocfg = O.UNetConfig(**SMALL)
        ref = O.MVUNetMotionModelRef(ocfg, n, F, hw).eval()
        O.init_synthetic_weights(ref, seed=0)
        model = MVUNetMotionModel(UNetConfig(**SMALL), ops=TorchRefOps(), num_views=n)
        model.load_state_dict(ref.state_dict())
        inp = O.synthetic_inputs(ocfg, videos, n, F, hw, seed=11, cfg_doubled=True)
        full = model(**inp).sample                     # unsharded, same process
        par = shard_unet(model)
        sharded = model(**inp).sample                  # default: gathers the attention input tokens (C wide)
        tok_bytes = par.gather_bytes
        par.gather_tokens = False                      # alternative: gathers the projected K|V (2C wide)
        par.gather_bytes = 0
        sharded_kv = model(**inp).sample
        err = max((sharded - full).abs().max().item(), (sharded_kv - full).abs().max().item())
        assert par.gather_bytes == 0 or tok_bytes < par.gather_bytes
        q.put((rank, err, par.cfg_shards, par.view_shards, par.gather_bytes, tuple(sharded.shape)))
    except Exception as e:   # surface the failure instead of letting the parent time out
        q.put((rank, repr(e), 0, 0, 0, ()))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,videos,expect", [(2, 2, 4, (2, 1)), (2, 2, 2, (1, 2)), (4, 2, 4, (2, 2)), (2, 4, 4, (1, 2)),
                                                   (8, 4, 8, (2, 4))])   # last: bench.py --gpus 8 (CFG x 4 view shards)
def test_sharded_forward_equals_unsharded(world, n, videos, expect):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 2, (8, 8), videos, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, cs, vs, gbytes, shape in res:
        assert not isinstance(err, str), err
        assert (cs, vs) == expect
        assert shape == (videos, 4, 2, 8, 8)
        assert err < 5e-4, (rank, err)   # fp32 summation-order noise through ~600 ops
        assert (gbytes > 0) == (vs > 1)


def test_layout_choice_and_local_videos():
    from animate3d_amd.parallel import ViewParallel
    assert ViewParallel.choose_layout(8, 2, 4) == (2, 4)      # BASELINE config 2 on 8 GPUs: CFG x views
    assert ViewParallel.choose_layout(4, 2, 4) == (2, 2)
    assert ViewParallel.choose_layout(2, 2, 4) == (2, 1)
    assert ViewParallel.choose_layout(8, 2, 8) == (2, 4)      # config 4: 8 views
    with pytest.raises(ValueError):
        ViewParallel.choose_layout(8, 1, 4)
