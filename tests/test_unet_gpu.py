"""GPU end-to-end parity of the MI355X UNet forward (HIP kernels, bf16) against

  (1) the CPU oracle (fp32 restatement of the reference forward, oracle/unet_ref.py), and
  (2) the same host logic executed with plain-torch ops that round to bf16 at the same points
      (tests/torch_ops.py) — separates "kernel is wrong" from "bf16 is coarse".

Real SD1.5 widths (320/640/1280/1280, head dims 40/80/160 — the only ones the attention kernels
implement) at a small latent, so the oracle finishes in seconds.

Stated tolerance (north_star: "within stated fp16/bf16 tolerance"): bf16 carries 8 significant bits
(rounding 2^-9 = 0.2 % per store) through ~600 dependent kernels per step;
bars: relative L2 error vs the fp32 oracle <= 3e-2 (observed 1.2-1.6e-2) with bf16 storage and <= 6e-3 (observed 1.9e-3) with fp16
storage.  The bf16-rounding emulation makes its own, independent rounding decisions (different summation orders inside its
ops), so HIP-vs-emulation, emulation-vs-oracle and HIP-vs-oracle are three samples of the same bf16 noise (all ~ 1.4e-2): what the
emulation leg asserts is that the HIP path is NOT WORSE than an honest bf16 execution of the same graph (<= 1.3 x its error) and
within sqrt(2) x the oracle bar of it.  The sharp check that the kernels compute the right function is the fp16-storage run of the
very same kernel sources, whose rounding noise is 8x smaller.
"""
import pytest
import torch

from animate3d_amd.config import UNetConfig
from animate3d_amd.unet import MVUNetMotionModel
from oracle import unet_ref as O
from tests.conftest import oracle_unet
from tests.torch_ops import TorchRefOps

pytestmark = pytest.mark.gpu

N_VIEWS, FRAMES, HW = 2, 3, (16, 16)


@pytest.fixture(scope="module")
def models():
    torch.manual_seed(0)
    ocfg = O.UNetConfig()
    ref = oracle_unet(ocfg, N_VIEWS, FRAMES, HW, seed=0)
    sd = ref.state_dict()
    hip = MVUNetMotionModel(UNetConfig(), num_views=N_VIEWS, device="cuda")
    missing, unexpected = hip.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    hip = hip.to(torch.bfloat16).eval()
    emu = MVUNetMotionModel(UNetConfig(), num_views=N_VIEWS, device="cuda", ops=TorchRefOps(torch.bfloat16, "cuda"))
    emu.load_state_dict(sd, strict=True)
    emu = emu.to(torch.bfloat16).eval()
    return ocfg, ref, hip, emu


def _cuda(inp):
    out = {}
    for k, v in inp.items():
        if torch.is_tensor(v):
            out[k] = v.cuda()
        elif isinstance(v, dict):
            out[k] = {kk: vv.cuda() for kk, vv in v.items()}
        else:
            out[k] = v
    return out


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item(), (a - b).abs().max().item(), b.abs().max().item()


@pytest.mark.parametrize("videos,seed,cond0", [(2, 1, False), (4, 2, False), (2, 3, True)])
def test_forward_parity(models, videos, seed, cond0):
    ocfg, ref, hip, emu = models
    inp = O.synthetic_inputs(ocfg, videos, N_VIEWS, FRAMES, HW, seed=seed, cfg_doubled=videos >= 2 * N_VIEWS)
    y_ref = ref(**inp, i2v_cond_time_zero=cond0).sample
    y_hip = hip(**_cuda(inp), i2v_cond_time_zero=cond0).sample
    y_emu = emu(**_cuda(inp), i2v_cond_time_zero=cond0).sample
    assert y_hip.shape == y_ref.shape and y_hip.dtype == torch.float32 and y_hip.is_cuda
    assert torch.isfinite(y_hip).all()
    e_or, mx_or, sc = _rel(y_hip, y_ref)
    e_em, mx_em, _ = _rel(y_hip, y_emu)
    e_emu_or, _, _ = _rel(y_emu, y_ref)
    print(f"[parity] unet V={videos} cond0={cond0}: hip-vs-oracle rel_l2={e_or:.3e} max_abs={mx_or:.3e} (|ref|max {sc:.3e}); "
          f"hip-vs-bf16emu rel_l2={e_em:.3e} max_abs={mx_em:.3e}; bf16emu-vs-oracle rel_l2={e_emu_or:.3e}")
    assert e_or <= 3e-2, f"HIP vs fp32 oracle: {e_or:.3e}"
    assert e_or <= 1.3 * e_emu_or, f"HIP ({e_or:.3e}) is worse than the bf16-rounding emulation of the same graph ({e_emu_or:.3e})"
    assert e_em <= 2.2e-2, f"HIP vs bf16-rounding emulation: {e_em:.3e}"


def test_forward_parity_config4_geometry(models):
    """BASELINE config 4's token geometry (8 views x 32 frames; motion_max_seq_length = 32 is the temporal limit) at an
    8x8 latent so the oracle finishes in seconds: 32-frame temporal attention and positional table, 8-view multi-view and
    first-frame row maps, 1x1-pixel feature maps at level 3."""
    ocfg, ref, hip, _ = models
    n, F, hw = 8, 32, (8, 8)
    ref8 = O.build_dense(ocfg, n, F, hw, state_dict=ref.state_dict())
    inp = O.synthetic_inputs(ocfg, n, n, F, hw, seed=5)
    y_ref = ref8(**inp).sample
    hip.num_views = n
    try:
        y_hip = hip(**_cuda(inp)).sample
    finally:
        hip.num_views = N_VIEWS
    assert y_hip.shape == y_ref.shape == (n, 4, F, 8, 8)
    e_or, mx_or, sc = _rel(y_hip, y_ref)
    print(f"[parity] unet 8 views x 32 frames: hip-vs-oracle rel_l2={e_or:.3e} max_abs={mx_or:.3e} (|ref|max {sc:.3e})")
    assert torch.isfinite(y_hip).all() and e_or <= 3e-2, f"HIP vs fp32 oracle: {e_or:.3e}"


SWITCH_SETS = [
    dict(mvdream_image_attn=False),                                        # a7: MVDream processor without the I2V branch
    dict(motion_use_alpha_blender=False),
    dict(motion_spatial_attn=False),                                       # vanilla AnimateDiff motion modules
    dict(motion_use_spatial_encoding=False),
    dict(motion_image_attn=True),                                          # 3-way SoftmaxAlphaBlender
    dict(motion_image_attn=True, motion_use_alpha_blender=False),
    dict(motion_image_attn=True, motion_spatial_attn=False),
    dict(motion_use_camera_encoding=True, motion_camera_encoding_type="sinusoid"),
    dict(motion_use_camera_encoding=True, motion_camera_encoding_type="learnable", motion_use_spatial_encoding=False),
    dict(motion_spatial_encoding_type="learnable"),
]


@pytest.mark.parametrize("kw", SWITCH_SETS, ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_switch_sets_on_the_hip_path(kw):
    """Every processor switch of inference.yaml:9-24 / attention_processor.py:478-540 that the released configuration leaves
    at its default, on the HIP kernels: two real-width levels (head dims 40 and 80, both with attention), against the fp32 oracle
    whose processors are pinned on the same switch sets by reference-generated vectors (tests/test_oracle_golden.py)."""
    arch = dict(block_out_channels=(320, 640), down_has_attn=(True, True), layers_per_block=1)
    n, F, hw, V = 2, 3, (16, 16), 2
    ocfg = O.UNetConfig(**arch, **kw)
    ref = O.build_dense(ocfg, n, F, hw, seed=0)
    hip = MVUNetMotionModel(UNetConfig(**arch, **kw), num_views=n, device="cuda")
    missing, unexpected = hip.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected
    hip = hip.to(torch.bfloat16).eval()
    inp = O.synthetic_inputs(ocfg, V, n, F, hw, seed=11)
    y_ref = ref(**inp).sample
    y = hip(**_cuda(inp)).sample
    e, mx, sc = _rel(y, y_ref)
    print(f"[parity] unet switch set {kw}: hip-vs-oracle rel_l2={e:.3e} max_abs={mx:.3e} (|ref|max {sc:.3e})")
    assert torch.isfinite(y).all() and e <= 3e-2


@pytest.mark.parametrize("hw", [(12, 20), (9, 10)])
def test_forced_upsample_sizes_on_the_hip_path(hw):
    """Latents that are not multiples of 8 (unet_motion_mv_model.py:690-698, 831-837): the cropped nearest-2x upsample of
    a3d_conv3x3_bf16 (up2x bits 1 / 2) on all four levels, against the oracle (whose forward is pinned on the 12 x 20 case by
    the reference's own forward, tests/golden/unet_forward.npz)."""
    n, F, V = 2, 2, 2
    ocfg = O.UNetConfig()
    ref = O.build_fast(ocfg, n, F, hw, seed=2)
    hip = MVUNetMotionModel(UNetConfig(), num_views=n, device="cuda")
    hip.load_state_dict(ref.state_dict(), strict=True)
    hip = hip.to(torch.bfloat16).eval()
    inp = O.synthetic_inputs(ocfg, V, n, F, hw, seed=12)
    y_ref = ref(**inp).sample
    y = hip(**_cuda(inp)).sample
    e, mx, sc = _rel(y, y_ref)
    print(f"[parity] unet {hw[0]}x{hw[1]} latent (forced upsample sizes): rel_l2={e:.3e} max_abs={mx:.3e} (|ref|max {sc:.3e})")
    assert y.shape == y_ref.shape and torch.isfinite(y).all() and e <= 3e-2


def test_output_dtype_and_determinism(models):
    ocfg, ref, hip, _ = models
    inp = _cuda(O.synthetic_inputs(ocfg, 2, N_VIEWS, FRAMES, HW, seed=5))
    inp["sample"] = inp["sample"].to(torch.bfloat16)
    a = hip(**inp).sample
    b = hip(**inp).sample
    assert a.dtype == torch.bfloat16 and torch.equal(a, b)


def test_fp16_model_and_inputs(models):
    """BASELINE configs 4/5 run the UNet in fp16 (animatemv_guidance.py:339-346 casts everything, incl. t, to fp16).  A model
    cast with ``.half()`` runs on the fp16-storage kernels (a3d_*_f16: v_mfma_f32_32x32x16_f16, fp32 accumulate) and returns
    fp16.  fp16 keeps 11 significant bits where bf16 keeps 8, so the bar against the fp32 oracle is 5x tighter than bf16's
    (6e-3 instead of 3e-2 relative L2); the same weights through the bf16 kernels are printed next to it."""
    ocfg, ref, hip, _ = models
    inp = O.synthetic_inputs(ocfg, 2, N_VIEWS, FRAMES, HW, seed=9)
    y_ref = ref(**inp).sample
    h16 = MVUNetMotionModel(UNetConfig(), num_views=N_VIEWS, device="cuda")
    h16.load_state_dict(ref.state_dict(), strict=True)
    h16 = h16.half().eval()
    assert h16.ops.act_dtype == torch.float16 and hip.ops.act_dtype == torch.bfloat16
    ci = _cuda(inp)
    ci["sample"] = ci["sample"].half()
    ci["encoder_hidden_states"] = ci["encoder_hidden_states"].half()
    ci["camera"] = ci["camera"].half()
    ci["added_cond_kwargs"] = {"image_embeds": ci["added_cond_kwargs"]["image_embeds"].half()}
    ci["timestep"] = torch.full((2,), 501.0, device="cuda", dtype=torch.float16)
    y = h16(**ci).sample
    assert y.dtype == torch.float16
    e, mx, sc = _rel(y, y_ref)
    e_bf, _, _ = _rel(hip(**_cuda(inp)).sample, y_ref)
    print(f"[parity] unet fp16 storage: rel_l2={e:.3e} max_abs={mx:.3e} (|ref|max {sc:.3e}); same weights, bf16 storage: {e_bf:.3e}")
    assert torch.isfinite(y).all() and e <= 6e-3
    # the op set follows the model: back to bf16 storage after .to(bfloat16)
    h16 = h16.to(torch.bfloat16)
    assert h16.ops.act_dtype == torch.bfloat16


def test_baseline_config1_shape():
    """BASELINE config 1 exactly: 1 view x 4 frames x 64x64 latent, no CFG (the reference's CPU-runnable case), against the oracle's
    output committed by tests/golden/make_gpu_tier_goldens.py (float16; weights and inputs re-drawn from the same seeds)."""
    import numpy as np
    from tests.golden import make_gpu_tier_goldens as G
    gold = np.load(G.OUT)
    ref = G.config1_weights()
    hip = MVUNetMotionModel(UNetConfig(), num_views=1, device="cuda")
    hip.load_state_dict(ref.state_dict(), strict=True)
    del ref
    hip = hip.to(torch.bfloat16).eval()
    inp = G.config1_inputs()
    assert abs(inp["sample"].double().sum().item() - float(gold["config1_in_checksum"])) < 1e-6, "inputs differ from the golden's"
    y_ref = torch.from_numpy(gold["config1_sample"].astype(np.float32))
    y = hip(**_cuda(inp), ).sample
    assert y.shape == y_ref.shape == (1, 4, 4, 64, 64)
    e, mx, sc = _rel(y, y_ref)
    print(f"[parity] unet config-1 shape (1v x 4f x 64x64) vs the oracle golden: rel_l2={e:.3e} max_abs={mx:.3e} (|ref|max {sc:.3e})")
    assert e <= 3e-2


def test_full_size_batch_independence():
    """BASELINE config 2 at full size (4 views x 16 frames x 64x64 latent, CFG-doubled V = 8): the oracle cannot run
    this in seconds, so check a size-independent property of the path instead — the two CFG halves never exchange
    data (every regrouping keeps b outermost), hence running the first half alone must reproduce it exactly."""
    from bench import make_inputs
    cfg = UNetConfig()
    m = MVUNetMotionModel(cfg, num_views=4, device="cuda")
    m.init_synthetic(seed=1)
    m = m.to(torch.bfloat16).eval()
    m.ops.split_k = False      # bit-for-bit across two batch sizes: split-K (mid-block convs) re-associates the K sum per launch shape
    inp = make_inputs(cfg, 8, 4, 16, (64, 64), torch.device("cuda"))
    full = m(**inp).sample
    half = dict(inp)
    half["sample"] = inp["sample"][:4]
    half["encoder_hidden_states"] = inp["encoder_hidden_states"][:4]
    half["camera"] = inp["camera"][:4]
    half["added_cond_kwargs"] = {"image_embeds": inp["added_cond_kwargs"]["image_embeds"][:4]}
    part = m(**half).sample
    assert torch.isfinite(full).all()
    assert torch.equal(full[:4], part), (full[:4] - part).abs().max().item()
    # frame 0 of the input is the clean conditioning frame: the prediction for it exists and is finite as well
    assert full.shape == (8, 4, 16, 64, 64)


@pytest.mark.parametrize("dtype,bar", [(torch.bfloat16, 3e-2), (torch.float16, 6e-3)])
def test_config2_full_size_against_the_oracle_golden(dtype, bar):
    """BASELINE config 2 at FULL size against the CPU oracle: 4 views x 16 frames x 64x64 latent (512^2 px), one CFG half (V = 4; the
    other half is independent of it bit for bit, test_full_size_batch_independence), level-0 multi-view attention over 16 384 keys —
    the launch shape that is half of the benchmark.  The oracle's output was computed once (tests/golden/make_config2_golden.py,
    ~45 min of CPU) and committed as fp16 (tests/golden/config2_full.npz); weights and inputs are re-drawn here from the same seeds.
    Bars: the end-to-end bars of the small shapes (bf16 3e-2, fp16 6e-3 relative L2; the golden's fp16 storage adds 3e-4)."""
    import os
    import numpy as np
    from tests.golden.make_config2_golden import FRAMES, HW, SEED_IN, SEED_W, VIEWS
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config2_full.npz"))
    want = torch.from_numpy(gold["sample"].astype(np.float32))
    ocfg = O.UNetConfig()
    ref = O.build_fast(ocfg, VIEWS, FRAMES, HW, seed=SEED_W)
    inp = O.synthetic_inputs(ocfg, VIEWS, VIEWS, FRAMES, HW, seed=SEED_IN)
    assert abs(inp["sample"].double().sum().item() - float(gold["in_checksum"])) < 1e-6 * max(1.0, abs(float(gold["in_checksum"])))
    hip = MVUNetMotionModel(UNetConfig(), num_views=VIEWS, device="cuda")
    hip.load_state_dict(ref.state_dict(), strict=True)
    del ref
    hip = hip.to(dtype).eval()
    y = hip(**_cuda(inp)).sample
    e, mx, sc = _rel(y, want)
    print(f"[parity] unet config 2 FULL size (4v x 16f x 64x64, {dtype}) vs the oracle golden: rel_l2={e:.3e} max_abs={mx:.3e} (|ref|max {sc:.3e})")
    assert y.shape == want.shape and torch.isfinite(y).all() and e <= bar


def test_config4_full_size_fp16():
    """BASELINE config 4 at FULL size on one GPU: 8 views x 32 frames x 64x64 latent, CFG-doubled V = 16, fp16 model and inputs
    (2.58 PFLOP per step, ~100 GB working set).  The oracle cannot run this; checked instead: finite output of the right shape,
    and the CFG halves do not exchange data (the first half alone reproduces its rows bit for bit)."""
    from bench import make_inputs
    cfg = UNetConfig()
    m = MVUNetMotionModel(cfg, num_views=8, device="cuda")
    m.init_synthetic(seed=1)
    m = m.half().eval()
    inp = make_inputs(cfg, 16, 8, 32, (64, 64), torch.device("cuda"))
    inp["sample"] = inp["sample"].half()
    full = m(**inp).sample
    assert full.shape == (16, 4, 32, 64, 64) and full.dtype == torch.float16 and torch.isfinite(full).all()
    half = dict(inp)
    half["sample"] = inp["sample"][:8]
    half["encoder_hidden_states"] = inp["encoder_hidden_states"][:8]
    half["camera"] = inp["camera"][:8]
    half["added_cond_kwargs"] = {"image_embeds": inp["added_cond_kwargs"]["image_embeds"][:8]}
    part = m(**half).sample
    assert torch.equal(full[:8], part), (full[:8].float() - part.float()).abs().max().item()


def test_loaded_native_library():
    """The .so that ran must be the in-tree one (no silent fallback)."""
    import os
    from animate3d_amd import hip_ops
    lib = hip_ops.load_library()
    assert lib.a3d_version().decode().startswith("animate3d_hip gfx950")
    with open("/proc/self/maps") as f:
        assert any(os.path.basename(hip_ops.lib_path()) in line for line in f)


def test_forward_is_hip_graph_capturable(models):
    """the C-ABI never allocates or synchronises and launches only on the caller's stream, so a whole denoise
    step can be captured into a HIP graph (launch-bound small configurations, e.g. the 4D-SDS call at a 32x32 latent, then
    replay ~1 700 kernels without any host work).  Capture one forward, replay it on new inputs, compare with eager."""
    ocfg, ref, hip, _ = models
    inp_a = _cuda(O.synthetic_inputs(ocfg, 2, N_VIEWS, FRAMES, HW, seed=21))
    inp_b = _cuda(O.synthetic_inputs(ocfg, 2, N_VIEWS, FRAMES, HW, seed=22))
    step = hip.capture_graph(**inp_a)
    for src in (inp_a, inp_b):
        y_static = step(**src).sample
        torch.cuda.synchronize()
        y_eager = hip(**src).sample
        assert torch.equal(y_static, y_eager), "graph replay differs from the eager forward"


def test_graph_capture_with_a_cu_reservation_active(models):
    """The CU reservation of the sharded path (parallel.py: compute-units left free for an overlapped RCCL collective) is a
    per-call argument of the C-ABI (``flags`` of a3d_gemm / a3d_conv3x3, include/animate3d_hip.h), held per op set — not process
    state: a capture taken while a reservation is active bakes ITS grid sizes into the graph, and neither a later change of the
    op set's value nor another op set in the same process can alter the replay.  The persistent kernels' results do not depend on
    the number of workgroups, so every combination must be bit-identical."""
    from animate3d_amd.hip_ops import HipOps
    ocfg, ref, hip, _ = models
    # 4 videos x 3 frames x 64x64 latent: 49 152 level-0 tokens = 192 row tiles, so the GEMMs and convolutions of levels 0 / 1 take
    # the persistent kernels, whose grid is what the reservation shrinks (the 16x16 latent of the other tests never reaches them)
    inp = _cuda(O.synthetic_inputs(ocfg, 2 * N_VIEWS, N_VIEWS, FRAMES, (64, 64), seed=23, cfg_doubled=True))
    y_plain = hip(**inp).sample.clone()
    other = HipOps()                                   # a second op set in the process keeps its own value
    assert other.reserved_cus == 0 and hip.ops.reserved_cus == 0
    hip.ops.reserved_cus = 24
    try:
        y_reserved = hip(**inp).sample.clone()
        step = hip.capture_graph(**inp)
        assert other.reserved_cus == 0
    finally:
        hip.ops.reserved_cus = 0
    y_replay = step(**inp).sample
    torch.cuda.synchronize()
    assert torch.equal(y_reserved, y_plain), "result depends on the number of reserved CUs"
    assert torch.equal(y_replay, y_plain), "graph captured under a reservation differs from the eager forward"


def test_controlnet_residual_inputs_on_the_gpu():
    """unet_motion_mv_model.py:787-796, 816-817 on the HIP path (a3d_axpby): a two-level real-width model (head dims 40 / 80) with
    additional residuals on every skip connection and on the mid block's output, against the oracle."""
    kw = dict(block_out_channels=(320, 640), down_has_attn=(True, False), layers_per_block=1)
    n, Fr, hw = 2, 2, (16, 16)
    ocfg = O.UNetConfig(**kw)
    ref = O.build_fast(ocfg, n, Fr, hw, seed=0)
    hip = MVUNetMotionModel(UNetConfig(**kw), num_views=n, device="cuda")
    hip.load_state_dict(ref.state_dict(), strict=True)
    hip = hip.half().eval()
    inp = O.synthetic_inputs(ocfg, n, n, Fr, hw, seed=4)
    B2 = n * Fr
    shapes = [(B2, 320, 16, 16), (B2, 320, 16, 16), (B2, 320, 8, 8), (B2, 640, 8, 8)]
    g = torch.Generator().manual_seed(2)
    down = [0.5 * torch.randn(s, generator=g) for s in shapes]
    mid = 0.5 * torch.randn((B2, 640, 8, 8), generator=g)
    y_ref = ref(**inp, down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
    y_plain = ref(**inp).sample
    y = hip(**_cuda(inp), down_block_additional_residuals=[t.cuda() for t in down], mid_block_additional_residual=mid.cuda()).sample
    e, mx, sc = _rel(y, y_ref)
    print(f"[parity] ControlNet residual inputs, fp16 storage vs oracle: rel_l2={e:.3e} (residuals move the output by {_rel(y_plain, y_ref)[0]:.3e})")
    assert e < 6e-3 and _rel(y_plain, y_ref)[0] > 10 * e


def _peaked(sd, gain):
    """A copy of the state dict with every query / key projection of the multi-view, first-frame and spatial motion attentions scaled by
    ``gain``: the seeded Kaiming-uniform weights give attention scores of sd ~0.33 nats (a nearly flat softmax, the max-free kernels never
    leave their fast path); gain 3 on both sides gives sd ~3 — the peaked softmax of trained MV-VDM weights."""
    out, hit = {}, 0
    for k, v in sd.items():
        t2d = k.endswith(("attn1.to_q.weight", "attn1.to_k.weight")) and "motion_modules" not in k
        if t2d or k.endswith(("to_q_i2v.weight", "to_q_sp.weight", "to_k_sp.weight")):
            v = v * gain
            hit += 1
        out[k] = v
    assert hit > 0
    return out


@pytest.mark.parametrize("dtype,bar", [(torch.bfloat16, 3e-2), (torch.float16, 6e-3)], ids=["bf16", "fp16"])
def test_forward_parity_on_a_peaked_softmax(dtype, bar):
    """VERDICT r5 weak spot 2: every other end-to-end test draws Kaiming weights (score sd ~0.3), so the spike / overflow / exact-re-run
    machinery of the LDS-DMA staged attention kernels was only ever covered at kernel level.  Here the q / k projections are scaled so
    that level-0 scores have sd ~3 (natural-log units), two real-width levels at a 32 x 32 latent so that BOTH staged kernels run
    (head_dim 40 over 2 048 keys, head_dim 80 over 512 keys), against the fp32 oracle with the same weights; bars unchanged.  The
    diagnostics counters (a3d_flash_attn_counted) say how many workgroups left the max-free fast path."""
    arch = dict(block_out_channels=(320, 640), down_has_attn=(True, True), layers_per_block=1)
    n, F, hw, V = 2, 2, (32, 32), 2
    ocfg = O.UNetConfig(**arch)
    base = O.build_dense(ocfg, n, F, hw, seed=0)
    sd = _peaked(base.state_dict(), 3.0)
    ref = O.build_dense(ocfg, n, F, hw, state_dict=sd)
    hip = MVUNetMotionModel(UNetConfig(**arch), num_views=n, device="cuda")
    missing, unexpected = hip.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    hip = hip.to(dtype).eval()
    inp = O.synthetic_inputs(ocfg, V, n, F, hw, seed=21)
    y_ref = ref(**inp).sample
    cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    hip.ops.attn_counters = cnt
    try:
        y = hip(**_cuda(inp)).sample
    finally:
        hip.ops.attn_counters = None
    voted, rerun, launched = (int(v) for v in cnt[:3].tolist())
    e, mx, sc = _rel(y, y_ref)
    # the same model on flat scores, as a reference point for the error and the counters
    hip0 = MVUNetMotionModel(UNetConfig(**arch), num_views=n, device="cuda")
    hip0.load_state_dict(base.state_dict(), strict=True)
    hip0 = hip0.to(dtype).eval()
    cnt0 = torch.zeros(4, dtype=torch.int32, device="cuda")
    hip0.ops.attn_counters = cnt0
    y0 = hip0(**_cuda(inp)).sample
    e0, _, _ = _rel(y0, base(**inp).sample)
    voted0, rerun0, launched0 = (int(v) for v in cnt0[:3].tolist())
    print(f"[parity] peaked softmax ({dtype}): hip-vs-oracle rel_l2={e:.3e} max_abs={mx:.3e} (|ref|max {sc:.3e}); staged-kernel workgroups {launched}, "
          f"sent exact by the spread vote {voted}, exact re-runs after overflow {rerun}; flat scores: rel_l2={e0:.3e}, {launched0} / {voted0} / {rerun0}")
    assert torch.isfinite(y).all() and e <= bar, f"HIP vs fp32 oracle on a peaked softmax: {e:.3e}"
    assert launched > 0 and launched == launched0, "the LDS-DMA staged attention kernels did not run"
    assert voted0 == 0 and rerun0 == 0, "flat scores must stay on the max-free fast path"
    if dtype == torch.bfloat16:
        assert voted == 0 and rerun == 0, "bf16 storage: sd 3 is far inside the 2^100 window, nothing may leave the fast path"
    else:
        # fp16 storage: P has a 20- to 28-binade window above the sample maximum (round 6: lifted with the sample's spread); at sd 3 a small
        # minority of workgroups may go exact (profiles/README.md, round 5: the OR-predictor sent EVERY workgroup there and nobody noticed
        # because the results stayed right)
        assert (voted + rerun) * 20 <= launched, f"{voted} + {rerun} of {launched} workgroups left the fast path at score sd 3"


@pytest.mark.parametrize("dtype,bar", [(torch.bfloat16, 3e-2), (torch.float16, 6e-3)], ids=["bf16", "fp16"])
def test_freeu_on_the_hip_path(dtype, bar):
    """enable_freeu (unet_motion_mv_model.py:562-585) on the kernels: two real-width levels (up block 0 at 8 x 8, up block 1 at 16 x 16), the
    FreeU repository's SD1.5 factors, against the oracle with diffusers' apply_freeu restated (oracle/freeu_ref.py; parity unpinned);
    switched off again the model reproduces its plain output bit for bit; a HIP-graph capture with FreeU on replays."""
    from oracle.freeu_ref import freeu
    arch = dict(block_out_channels=(320, 640), down_has_attn=(True, True), layers_per_block=1)
    n, F, hw, V = 2, 2, (16, 16), 2
    ocfg = O.UNetConfig(**arch)
    ref = O.build_dense(ocfg, n, F, hw, seed=0)
    hip = MVUNetMotionModel(UNetConfig(**arch), num_views=n, device="cuda")
    hip.load_state_dict(ref.state_dict(), strict=True)
    hip = hip.to(dtype).eval()
    inp = O.synthetic_inputs(ocfg, V, n, F, hw, seed=31)
    cinp = _cuda(inp)
    y_plain = hip(**cinp).sample
    factors = dict(s1=0.9, s2=0.2, b1=1.5, b2=1.6)
    with freeu(ref, **factors):
        y_ref = ref(**inp).sample
    hip.enable_freeu(**factors)
    y = hip(**cinp).sample
    e, mx, sc = _rel(y, y_ref)
    moved = _rel(y, y_plain)[0]
    print(f"[parity] FreeU ({dtype}): hip-vs-oracle rel_l2={e:.3e} max_abs={mx:.3e} (|ref|max {sc:.3e}); FreeU moves the output by {moved:.3e}")
    assert torch.isfinite(y).all() and e <= bar and moved > 5 * e
    step = hip.capture_graph(**cinp)
    assert torch.equal(step(**cinp).sample, y)
    hip.disable_freeu()
    assert torch.equal(hip(**cinp).sample, y_plain)
