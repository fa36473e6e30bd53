"""CPU tests of the product's HOST logic: animate3d_amd/unet.py is executed with the plain-torch
op set of tests/torch_ops.py (fp32) and must reproduce the oracle's NCHW restatement of the reference
forward.  This checks the NHWC/token-major layout, the row maps that replace the reference's
rearranges, fused-QKV / conv weight packing, once-per-video text/IP K/V, epilogue fusions and the
state-dict key compatibility — everything except the HIP kernels themselves (those are -m gpu)."""
import numpy as np
import pytest
import torch

from animate3d_amd.config import UNetConfig
from animate3d_amd.embeddings import get_camera, sine_pos_2d
from animate3d_amd.unet import MVUNetMotionModel
from oracle import unet_ref as O
from tests.torch_ops import TorchRefOps

SMALL = dict(block_out_channels=(32, 64, 64, 64))


def _pair(n, F, hw, dense=True, **cfgkw):
    ocfg = O.UNetConfig(**SMALL, **cfgkw)
    ref = O.MVUNetMotionModelRef(ocfg, n, F, hw).eval()
    O.init_synthetic_weights(ref, seed=0, dense=dense)
    model = MVUNetMotionModel(UNetConfig(**SMALL, **cfgkw), ops=TorchRefOps(), num_views=n)
    missing, unexpected = model.load_state_dict(ref.state_dict(), strict=True)
    assert not missing and not unexpected
    return ocfg, ref, model


def test_state_dict_keys_match_oracle_exactly():
    ocfg, ref, model = _pair(2, 3, (8, 8))
    assert list(model.state_dict().keys()) == list(ref.state_dict().keys()) or \
        set(model.state_dict().keys()) == set(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert model.state_dict()[k].shape == v.shape, k


def test_missing_key_count_motion_only_checkpoint():
    """inference.py:214-223: loading a motion-modules-only checkpoint must report 726 missing keys on the
    full-width model (SURVEY.md §8c KAT 1).  Checked on key names only (meta device, no memory)."""
    model = MVUNetMotionModel(UNetConfig(), device="meta")
    keys = list(model.state_dict().keys())
    saved = [k for k in keys if "i2v." in k or "motion_modules." in k]       # train.yaml:34-36 trainable subset
    assert len(keys) - len(saved) == 726
    assert len(model.attn_processors) == 16 * 2 + 21 * 2


@pytest.mark.parametrize("n,F,hw,videos", [(2, 3, (8, 8), 2), (1, 4, (8, 16), 1), (2, 2, (16, 8), 4)])
def test_forward_matches_oracle(n, F, hw, videos):
    ocfg, ref, model = _pair(n, F, hw)
    inp = O.synthetic_inputs(ocfg, videos, n, F, hw, seed=3, cfg_doubled=videos >= 2 * n)
    y_ref = ref(**inp).sample
    y = model(**inp).sample
    assert y.shape == y_ref.shape == (videos, 4, F, hw[0], hw[1])
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=2e-3, atol=2e-4)


def test_forward_cond_time_zero_and_tuple_return():
    ocfg, ref, model = _pair(2, 3, (8, 8))
    inp = O.synthetic_inputs(ocfg, 2, 2, 3, (8, 8), seed=5)
    y_ref = ref(**inp, i2v_cond_time_zero=True).sample
    (y,) = model(**inp, i2v_cond_time_zero=True, return_dict=False)
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=2e-3, atol=2e-4)
    assert not np.allclose(y.numpy(), model(**inp).sample.numpy(), atol=1e-5)


@pytest.mark.parametrize("hw", [(12, 20), (9, 10), (8, 15)])
def test_forced_upsample_sizes(hw):
    """unet_motion_mv_model.py:690-698, 831-837: latents whose sides are not multiples of 8 — every upsampler is told the
    size of the skip it has to meet (nearest interpolation to 2h - 1 = the 2x upsample minus its last row)."""
    ocfg, ref, model = _pair(2, 2, hw)
    inp = O.synthetic_inputs(ocfg, 2, 2, 2, hw, seed=9)
    y_ref = ref(**inp).sample
    y = model(**inp).sample
    assert y.shape == y_ref.shape == (2, 4, 2, hw[0], hw[1])
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=2e-3, atol=2e-4)


UNRELEASED_SWITCH_SETS = [
    dict(motion_image_attn=True),                                          # 3-way SoftmaxAlphaBlender
    dict(motion_image_attn=True, motion_use_alpha_blender=False),          # plain sum of three branches
    dict(motion_image_attn=True, motion_spatial_attn=False),               # temporal + image, AlphaBlender(image, temporal)
    dict(motion_use_camera_encoding=True, motion_camera_encoding_type="sinusoid"),
    dict(motion_use_camera_encoding=True, motion_camera_encoding_type="learnable", motion_use_spatial_encoding=False),
    dict(motion_spatial_encoding_type="learnable"),
    dict(motion_use_spatial_encoding=False),                               # spatial attention, block-level temporal encoding
    dict(motion_spatial_attn=False),                                       # vanilla AnimateDiff motion modules
]


@pytest.mark.parametrize("kw", UNRELEASED_SWITCH_SETS, ids=lambda kw: ",".join(f"{k[7:]}={v}" for k, v in kw.items()))
def test_unreleased_motion_switch_sets(kw):
    """attention_processor.py:514-535, 565-580, 672-713, 727-744; embeddings.py:99-157 (SURVEY §8 a9 beyond the released
    configuration): the product's host logic against the oracle, whose processor is pinned on the same switch sets."""
    ocfg, ref, model = _pair(2, 2, (8, 8), **kw)
    inp = O.synthetic_inputs(ocfg, 2, 2, 2, (8, 8), seed=7)
    np.testing.assert_allclose(model(**inp).sample.numpy(), ref(**inp).sample.numpy(), rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("kw", [dict(mvdream_image_attn=False), dict(motion_use_alpha_blender=False)])
def test_processor_switches(kw):
    ocfg, ref, model = _pair(2, 2, (8, 8), **kw)
    inp = O.synthetic_inputs(ocfg, 2, 2, 2, (8, 8), seed=7)
    np.testing.assert_allclose(model(**inp).sample.numpy(), ref(**inp).sample.numpy(), rtol=2e-3, atol=2e-4)


def test_reference_error_behaviour():
    ocfg, ref, model = _pair(2, 2, (8, 8))
    inp = O.synthetic_inputs(ocfg, 2, 2, 2, (8, 8))
    with pytest.raises(AssertionError):
        model(**{**inp, "num_views": 3})
    bad = dict(inp); bad["added_cond_kwargs"] = {}
    with pytest.raises(ValueError):
        model(**bad)
    with pytest.raises(AssertionError):
        model(**{**inp, "camera": inp["camera"][:1]})


def test_set_attn_processor_contract():
    _, _, model = _pair(2, 2, (8, 8))
    procs = model.attn_processors
    assert all(k.endswith(".processor") for k in procs)
    with pytest.raises(ValueError):
        model.set_attn_processor({k: v for k, v in list(procs.items())[:-1]})
    model.set_attn_processor(procs)        # round trip


def test_tables_against_reference_goldens(golden):
    C = int(golden["meta"][4])
    for key in ("4x4", "3x5"):
        h, w = [int(v) for v in key.split("x")]
        ref = torch.from_numpy(golden[f"sine2d/{key}"]).permute(1, 2, 0).reshape(h * w, C)
        np.testing.assert_allclose(sine_pos_2d(C // 2, h, w).numpy(), ref.numpy(), rtol=1e-5, atol=2e-6)
    for nv in (4, 8):
        np.testing.assert_allclose(get_camera(nv).numpy(), golden[f"camera/{nv}"], rtol=1e-5, atol=1e-6)


def _zero_(model, pattern):
    n = 0
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if pattern(k):
                v.zero_(); n += 1
    assert n > 0
    return n


def test_structural_kat_frames_decouple_without_motion_and_i2v():
    """SURVEY.md §8c KAT (3): with every motion module's proj_out zeroed (and to_out_i2v = 0, its initial value,
    inference.py:161-165) nothing couples the frames of a video any more, so the F-frame forward must equal the
    single-frame forwards frame by frame — on the product (token-major layout, row maps) and on the oracle."""
    n, F, hw = 2, 3, (8, 8)
    ocfg, ref, model = _pair(n, F, hw)
    pat = lambda k: ("motion_modules" in k and ".proj_out." in k) or "to_out_i2v" in k
    _zero_(ref, pat)
    model.load_state_dict(ref.state_dict(), strict=True)
    ref1 = O.MVUNetMotionModelRef(ocfg, n, 1, hw).eval()
    ref1.load_state_dict(ref.state_dict(), strict=True)
    inp = O.synthetic_inputs(ocfg, n, n, F, hw, seed=11)
    y_full, yr_full = model(**inp).sample, ref(**inp).sample
    for f in range(F):
        one = dict(inp, sample=inp["sample"][:, :, f:f + 1].contiguous())
        np.testing.assert_allclose(model(**one).sample[:, :, 0].numpy(), y_full[:, :, f].numpy(), rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(ref1(**one).sample[:, :, 0].numpy(), yr_full[:, :, f].numpy(), rtol=2e-3, atol=2e-4)


def test_structural_kat_alpha_zero_is_vanilla_motion_module():
    """SURVEY.md §8c KAT (2): mix_factor -> -inf (alpha = sigmoid = 0) removes the multi-view spatial branch of every
    SpatioTemporalI2V processor (attention_processor.py:708-709), i.e. equals the model built without that branch."""
    n, F, hw = 2, 2, (8, 8)
    ocfg, ref, model = _pair(n, F, hw)
    sd = {k: (torch.full_like(v, -1e4) if k.endswith("mix_factor") else v) for k, v in ref.state_dict().items()}
    assert any(k.endswith("mix_factor") for k in sd)
    model.load_state_dict(sd, strict=True)
    ocfg2, ref2, model2 = _pair(n, F, hw, motion_spatial_attn=False)
    model2.load_state_dict({k: v for k, v in sd.items() if k in model2.state_dict()}, strict=True)
    inp = O.synthetic_inputs(ocfg, n, n, F, hw, seed=13)
    np.testing.assert_allclose(model(**inp).sample.numpy(), model2(**inp).sample.numpy(), rtol=2e-3, atol=2e-4)


def test_from_unet2d_copies_what_the_reference_copies():
    """Weight mapping of ``from_unet2d`` / ``load_motion_modules`` (unet_motion_mv_model.py:275-368, 394-402): the fixture
    tests/golden/from_unet2d.json is the provenance of every parameter after the REFERENCE's own methods ran on three differently
    seeded module trees (tests/golden/make_from_unet2d_goldens.py); the drop-in must take the same tensors from the same source
    and leave the same ones (``encoder_hid_proj``) alone."""
    import json
    import os
    from tests.torch_ops import TorchRefOps
    tags = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "from_unet2d.json")))
    small = dict(block_out_channels=(32, 64, 64, 64), num_attention_heads=4, norm_num_groups=8)

    def tree(seed):
        m = O.MVUNetMotionModelRef(O.UNetConfig(**small), 2, 3, (8, 8)).eval()
        return O.init_synthetic_weights(m, seed=seed, dense=True)

    unet2d, adapter = tree(1), tree(2)
    model = MVUNetMotionModel.from_unet2d(unet2d, adapter, config=UNetConfig(**small), ops=TorchRefOps(), num_views=2)
    got, a, b = model.state_dict(), unet2d.state_dict(), adapter.state_dict()
    assert sorted(got.keys()) == sorted(tags.keys())
    counts = {}
    for k, tag in tags.items():
        counts[tag] = counts.get(tag, 0) + 1
        if tag == "unet":
            assert torch.equal(got[k], a[k]), k
        elif tag == "adapter":
            assert torch.equal(got[k], b[k]), k
        elif tag == "untouched":
            assert not torch.equal(got[k], a[k]) and not torch.equal(got[k], b[k]), k
    assert counts["unet"] == 768 and counts["adapter"] == 798 and counts["untouched"] == 4
    # a plain 2-D UNet has no I2V keys: the branch is then initialised from the LOADED to_q (inference.py:161-165), not from
    # the construction-time random one
    class Plain:
        def state_dict(self):
            return {k: v for k, v in a.items() if "i2v" not in k}
    m2 = MVUNetMotionModel.from_unet2d(Plain(), adapter, config=UNetConfig(**small), ops=TorchRefOps(), num_views=2).state_dict()
    pre = "down_blocks.0.attentions.0.transformer_blocks.0.attn1."
    assert torch.equal(m2[pre + "processor.to_q_i2v.weight"], a[pre + "to_q.weight"])
    # load_weights=False builds the topology only
    bare = MVUNetMotionModel.from_unet2d(unet2d, adapter, load_weights=False, config=UNetConfig(**small), ops=TorchRefOps(), num_views=2)
    assert not torch.equal(bare.state_dict()["conv_in.weight"], a["conv_in.weight"])


def test_processor_installation_matches_the_reference_statements():
    """Which processor every attention layer gets, with which sizes (inference.py:90-192): the fixture
    tests/golden/processor_install.json records what the REFERENCE's own statements decided for the released switch set
    (tests/golden/make_processor_install_goldens.py).  The oracle's and the product's role-based installation must agree layer by
    layer; at the reference's hard-coded 256-px sample size the oracle's feature sizes are the reference's."""
    import json
    import os
    from animate3d_amd import modules as M
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "processor_install.json")))
    n, f = fx["num_views"], fx["num_frames"]
    want = fx["processors"]
    with torch.device("meta"):
        ref = O.MVUNetMotionModelRef(O.UNetConfig(), n, f, (32, 32))             # 256 px / 8 = inference.py:93-95
    hip = MVUNetMotionModel(UNetConfig(), ops=object(), num_views=n, device="meta")
    oracle_procs = {f"{name}.processor": m.processor for name, m in ref.named_modules() if isinstance(m, O.Attention)}
    product_procs = hip.attn_processors
    assert sorted(oracle_procs) == sorted(product_procs) == sorted(want)
    # traversal order of the reference's own ``attn_processors`` (unet_motion_mv_model.py:439-462) over these module trees
    # (down_blocks, up_blocks, mid_block: the registration order of unet_motion_mv_model.py:152-153,187, which is also the order
    #  in which diffusers' IP-Adapter loader numbers the attn2 layers 1, 3, .., 31 — so the mid block is number 31)
    assert list(product_procs.keys()) == fx["attn_processor_order"]
    attn2 = [k for k in fx["attn_processor_order"] if "motion_modules" not in k and k.endswith("attn2.processor")]
    assert len(attn2) == 16 and attn2[-1].startswith("mid_block") and attn2[6].startswith("up_blocks.1.attentions.0")
    classes = {"SpatioTemporalI2VXFormersAttnProcessor": (O.SpatioTemporalProc, M.SpatioTemporalI2VAttnProcessor),
               "MVDreamI2VXFormersAttnProcessor": (O.MVDreamI2VProc, M.MVDreamI2VAttnProcessor),
               "IPAdapterXFormersAttnProcessor": (O.IPAdapterProc, M.IPAdapterAttnProcessor)}
    for name, rec in want.items():
        oc, pc = classes[rec["class"]]
        op, pp = oracle_procs[name], product_procs[name]
        assert type(op) is oc and type(pp) is pc, name
        if "hidden_size" in rec:
            assert pp.hidden_size == rec["hidden_size"], name
            assert getattr(op, "hidden_size", rec["hidden_size"]) == rec["hidden_size"], name
        if rec["class"].startswith("SpatioTemporal"):
            assert op.feature_hw == (rec["feature_size"], rec["feature_size"]) and (op.num_views, op.num_frames) == (n, f), name
            assert pp.use_alpha_blender == rec["use_alpha_blender"] and hasattr(pp, "alpha_blender") and hasattr(pp, "time_pos_embed")
        if rec["class"].startswith("MVDreamI2V"):
            assert rec["to_q_i2v_is_to_q"] and rec["to_out_i2v_zero"]
            assert list(pp.to_out_i2v.weight.shape) == rec["to_out_i2v_shape"] == list(op.to_out_i2v.weight.shape)
        if rec["class"].startswith("IPAdapter"):
            assert pp.cross_attention_dim == rec["cross_attention_dim"] and list(pp.num_tokens) == list(rec["num_tokens"])
    # inference.py:176-192: every motion module's BasicTransformerBlock loses its pos_embed
    assert len(fx["pos_embed_none"]) == 21
    for path in fx["pos_embed_none"]:
        for model in (ref, hip):
            mod = model
            for part in path.split("."):
                mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
            assert mod.transformer_blocks[0].pos_embed is None, path
    # the I2V initialisation itself (to_q_i2v := to_q, to_out_i2v := 0) on real (small) modules
    small = dict(block_out_channels=(32, 64, 64, 64), num_attention_heads=4, norm_num_groups=8)
    live = O.MVUNetMotionModelRef(O.UNetConfig(**small), 2, 2, (8, 8))
    blk = live.down_blocks[0].attentions[0].transformer_blocks[0]
    assert torch.equal(blk.attn1.processor.to_q_i2v.weight, blk.attn1.to_q.weight) and float(blk.attn1.processor.to_out_i2v.weight.detach().abs().max()) == 0.0
    prod = MVUNetMotionModel(UNetConfig(**small), ops=object(), num_views=2)
    pb = prod.down_blocks[0].attentions[0].transformer_blocks[0]
    assert torch.equal(pb.attn1.processor.to_q_i2v.weight, pb.attn1.to_q.weight) and float(pb.attn1.processor.to_out_i2v.weight.detach().abs().max()) == 0.0
    assert float(pb.attn1.processor.to_out_i2v.bias.detach().abs().max()) == 0.0


@pytest.mark.parametrize("switches", [dict(mvdream_image_attn=False), dict(motion_use_alpha_blender=False), dict(motion_spatial_attn=False),
                                      dict(motion_use_spatial_encoding=False), dict(motion_spatial_attn=False, mvdream_image_attn=False)])
def test_non_released_switch_sets_match_oracle(switches):
    """The processor switches the released configs never flip (configs/*/​*.yaml all use spatial attention + sinusoid encoding +
    alpha blender + I2V): product host logic == oracle there too.  In particular where diffusers' BasicTransformerBlock keeps its
    temporal ``pos_embed`` (inference.py:176-178 nulls it only when the spatial branch carries an encoding): spatial branch off ->
    PE on the temporal attention's input; spatial on without encoding -> PE on the input of BOTH branches."""
    small = dict(block_out_channels=(32, 64, 64, 64), num_attention_heads=4, norm_num_groups=8, **switches)
    n, videos, F, hw = 2, 4, 3, (8, 8)
    ref = O.MVUNetMotionModelRef(O.UNetConfig(**small), n, F, hw).eval()
    O.init_synthetic_weights(ref, seed=1)
    model = MVUNetMotionModel(UNetConfig(**small), ops=TorchRefOps(), num_views=n)
    model.load_state_dict(ref.state_dict())
    inp = O.synthetic_inputs(O.UNetConfig(**small), videos, n, F, hw, seed=3, cfg_doubled=True)
    inp["timestep"] = torch.tensor([7, 501, 999, 250])
    want = ref(**inp, i2v_cond_time_zero=True).sample
    got = model(**inp, i2v_cond_time_zero=True).sample
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-3, atol=2e-4)
    if not (small.get("motion_spatial_attn", True) and small.get("motion_use_spatial_encoding", True)):
        blk = ref.down_blocks[0].motion_modules[0].transformer_blocks[0]
        assert blk.pos_embed is not None                                  # the kept temporal PE matters: dropping it changes the result
        saved = [b.transformer_blocks[0].pos_embed for b in ref.modules() if isinstance(b, O.TransformerTemporalModel)]
        for b in ref.modules():
            if isinstance(b, O.TransformerTemporalModel):
                b.transformer_blocks[0].pos_embed = None
        assert (ref(**inp, i2v_cond_time_zero=True).sample - want).abs().max() > 1e-3


def test_structural_config_variants_match_oracle():
    """Topology knobs away from the SD1.5 values (block count and widths, attention placement, heads, layers per block, token
    counts, IP scale): product host logic == oracle; unsupported ones raise at construction."""
    base = dict(block_out_channels=(32, 64, 64, 64), num_attention_heads=4, norm_num_groups=8)
    for sw in (dict(ip_scale=0.4, ip_num_tokens=8), dict(layers_per_block=1, down_has_attn=(True, False, True, False)),
               dict(block_out_channels=(32, 64), down_has_attn=(True, False), motion_num_attention_heads=2),
               dict(cross_attention_dim=48, motion_max_seq_length=8, out_channels=8, ip_image_embed_dim=40)):
        kw = dict(base, **sw)
        n, videos, F, hw = 2, 4, 3, (8, 8)
        ocfg = O.UNetConfig(**kw)
        ref = O.MVUNetMotionModelRef(ocfg, n, F, hw).eval()
        O.init_synthetic_weights(ref, seed=1)
        model = MVUNetMotionModel(UNetConfig(**kw), ops=TorchRefOps(), num_views=n)
        model.load_state_dict(ref.state_dict(), strict=True)
        inp = O.synthetic_inputs(ocfg, videos, n, F, hw, seed=3, cfg_doubled=True)
        np.testing.assert_allclose(model(**inp).sample.numpy(), ref(**inp).sample.numpy(), rtol=2e-3, atol=2e-4, err_msg=str(sw))
    with pytest.raises(ValueError):
        MVUNetMotionModel(UNetConfig(**dict(base, in_channels=9)), ops=TorchRefOps(), num_views=2)


def _residual_shapes(cfg, B2, hw):
    """Shapes of down_block_res_samples (unet_motion_mv_model.py:769-785): conv_in, then per down block its resnets' outputs and the
    down-sampler's output; plus the mid block's output."""
    h, w = hw
    boc = cfg.block_out_channels
    shapes = [(B2, boc[0], h, w)]
    for i, c in enumerate(boc):
        shapes += [(B2, c, h, w)] * cfg.layers_per_block
        if i != len(boc) - 1:
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            shapes.append((B2, c, h, w))
    return shapes, (B2, boc[-1], h, w)




@pytest.mark.parametrize("hw", [(16, 16), (12, 20)])
def test_freeu_matches_the_restated_diffusers_arithmetic(hw):
    """enable_freeu (unet_motion_mv_model.py:562-585) stores four factors on the up blocks; diffusers' up blocks then run apply_freeu in front
    of every skip concatenation of up blocks 0 and 1 (backbone half x b, low frequencies of the skip x s).  Product (token rows, rocFFT / torch.fft
    over the H, W axes of the NHWC view) against the oracle's NCHW restatement (oracle/freeu_ref.py; third-party arithmetic: parity unpinned),
    power-of-two and other plane sizes (12 x 20 also takes the forced-upsample-size path), and: off again = bit-identical to never on."""
    from oracle.freeu_ref import freeu
    ocfg, ref, model = _pair(2, 2, hw)
    inp = O.synthetic_inputs(ocfg, 2, 2, 2, hw, seed=9)
    y_off = model(**inp).sample
    y_plain = ref(**inp).sample
    factors = dict(s1=0.9, s2=0.2, b1=1.5, b2=1.6)             # the FreeU repository's SD1.5 values
    with freeu(ref, **factors):
        y_ref = ref(**inp).sample
    assert torch.equal(ref(**inp).sample, y_plain)              # the context manager restores the oracle
    model.enable_freeu(**factors)
    y = model(**inp).sample
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=2e-3, atol=2e-4)
    assert (y_ref - y_plain).abs().max() > 1e-2 * y_plain.abs().max()      # it does something
    model.disable_freeu()
    assert torch.equal(model(**inp).sample, y_off)


@pytest.mark.parametrize("hw", [(8, 8), (12, 20)])
def test_controlnet_residual_inputs(hw):
    """unet_motion_mv_model.py:787-796, 816-817: additional residuals on the skip connections and on the mid block's output
    (given as the reference gives them: [(V F), C, h, w]); and the LoRA `scale` of cross_attention_kwargs, a no-op without LoRA layers."""
    n, Fr = 2, 2
    ocfg, ref, model = _pair(n, Fr, hw)
    inp = O.synthetic_inputs(ocfg, n, n, Fr, hw, seed=9)
    down_shapes, mid_shape = _residual_shapes(ocfg, n * Fr, hw)
    g = torch.Generator().manual_seed(4)
    down = tuple(0.3 * torch.randn(s, generator=g) for s in down_shapes)
    mid = 0.3 * torch.randn(mid_shape, generator=g)
    keep = [t.clone() for t in down] + [mid.clone()]
    y = model(**inp, down_block_additional_residuals=down, mid_block_additional_residual=mid, cross_attention_kwargs={"scale": 0.8}).sample
    assert all(torch.equal(a, b) for a, b in zip(list(down) + [mid], keep)), "the call must not write into the caller's residual tensors"
    y_ref = ref(**inp, down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=2e-3, atol=2e-4)
    assert not np.allclose(y.numpy(), model(**inp).sample.numpy(), atol=1e-3)
    np.testing.assert_allclose(model(**inp, mid_block_additional_residual=mid).sample.numpy(),
                               ref(**inp, mid_block_additional_residual=mid).sample.numpy(), rtol=2e-3, atol=2e-4)
    with pytest.raises(ValueError):
        model(**inp, down_block_additional_residuals=down[:-1])
    with pytest.raises(ValueError):
        model(**inp, timestep_cond=torch.zeros(n, 4))
    with pytest.raises(ValueError, match="inconsistent with the attention shapes"):
        model(**inp, attention_mask=torch.ones(n, 77))
    with pytest.raises(NotImplementedError):
        model(**inp, cross_attention_kwargs={"gligen": {}})


def test_saved_motion_modules_load_into_a_stock_motion_adapter(tmp_path):
    """The directory ``save_motion_modules`` writes, read back by diffusers' own ``MotionAdapter.from_pretrained`` (what the reference's
    ``save_pretrained`` output is consumed by).  diffusers is not part of the offline image this repo is built in: the test runs
    wherever it is installed and is skipped otherwise — until then the claim in ``save_motion_modules``' docstring stays unverified."""
    diffusers = pytest.importorskip("diffusers")
    ocfg, ref, model = _pair(2, 2, (8, 8))
    model.save_motion_modules(str(tmp_path / "adapter"))
    adapter = diffusers.MotionAdapter.from_pretrained(str(tmp_path / "adapter"))
    got = adapter.state_dict()
    want = {k: v for k, v in model.state_dict().items() if model._MOTION_KEY.match(k) and ".processor." not in k}
    assert set(want) <= set(got)
    for k, v in want.items():
        assert torch.equal(got[k].float(), v.detach().cpu().float()), k


def test_motion_module_save_load_freeze_surface(tmp_path):
    """unet_motion_mv_model.py:370-438 and the diffusers conveniences the class inherits: freeze_unet2d_params, save_motion_modules ->
    a MotionAdapter directory -> load_motion_modules, and the no-op / raising helpers."""
    import json
    from types import SimpleNamespace
    from safetensors.torch import load_file
    ocfg, ref, model = _pair(2, 2, (8, 8))
    model.freeze_unet2d_params()
    trainable = [k for k, p in model.named_parameters() if p.requires_grad]
    assert trainable and all("motion_modules." in k for k in trainable)
    assert sum("motion_modules." in k for k, _ in model.named_parameters()) == len(trainable)
    model.save_motion_modules(str(tmp_path / "adapter"))
    meta = json.load(open(tmp_path / "adapter" / "config.json"))
    assert meta["_class_name"] == "MotionAdapter" and meta["block_out_channels"] == list(ocfg.block_out_channels)
    sd = load_file(str(tmp_path / "adapter" / "diffusion_pytorch_model.safetensors"))
    assert sd and all("motion_modules." in k and ".processor." not in k for k in sd)
    other = MVUNetMotionModel(UNetConfig(**SMALL), ops=TorchRefOps(), num_views=2)
    before = other.state_dict()["down_blocks.0.resnets.0.conv1.weight"].clone()
    other.load_motion_modules(SimpleNamespace(state_dict=lambda: sd))
    pes = [k for k in sd if k.endswith(".pos_embed.pe")]       # the buffers a stock diffusers MotionAdapter registers: one per temporal transformer block
    assert len(pes) == sum(k.endswith(".transformer_blocks.0.attn1.to_q.weight") for k in sd) and meta["motion_max_seq_length"] == sd[pes[0]].shape[1]
    assert all(sd[k].shape == (1, meta["motion_max_seq_length"], sd[k[:-len("pos_embed.pe")] + "attn1.to_q.weight"].shape[0]) for k in pes)
    for k, v in sd.items():
        if k not in pes:                                         # (this model computes the encodings on the fly: no such buffer to load into)
            assert torch.equal(other.state_dict()[k], v), k
    assert torch.equal(other.state_dict()["down_blocks.0.resnets.0.conv1.weight"], before)       # nothing else touched
    with pytest.raises(KeyError):
        other.load_motion_modules(SimpleNamespace(state_dict=lambda: {k: v for k, v in sd.items() if "down_blocks.1" not in k}))
    other.fuse_qkv_projections(); other.unfuse_qkv_projections(); other.enable_forward_chunking(2, 1); other.disable_forward_chunking(); other.disable_freeu()
    other.enable_freeu(0.9, 0.2, 1.2, 1.4); other.disable_freeu()          # (arithmetic: test_freeu_matches_the_restated_diffusers_arithmetic)
    assert other._freeu_factors is None
    with pytest.raises(ValueError):
        other.set_default_attn_processor()
    with pytest.raises(ValueError):
        other.enable_forward_chunking(2, 3)


def test_attention_mask_cannot_be_consistent_in_the_reference_either():
    """``attention_mask`` (unet_motion_mv_model.py:639, 700-703): the shape walk of what the reference builds from it
    (oracle.unet_ref.reference_attention_mask_trace: regrouping of attention_processor.py:340, diffusers' prepare_attention_mask, the
    xformers bias contract) is inconsistent at some resolution for EVERY mask shape a caller could pass to the full UNet — the reference
    raises inside xformers, the product raises with the same walk in its message."""
    from oracle.unet_ref import reference_attention_mask_trace as trace
    V, n, F, heads = 8, 4, 16, 8
    res = [64 * 64, 32 * 32, 16 * 16, 8 * 8]                       # tokens per image at the four resolutions that attend (config 2)
    cands = set()
    for B in (1, V // n, V, (V // n) * F, V * F, (V // n) * F * heads, V * F * heads):
        for K in (77, 81, *res, *(n * l for l in res)):
            cands.add((B, K))
    for shape in sorted(cands):
        walk = trace(shape, V, n, F, heads, res)
        assert not all(ok for *_, ok in walk), (shape, walk)
        assert sum(ok for *_, ok in walk) <= 1                     # at most one resolution can match
    # the only consistent case: a UNet that attends at ONE resolution and a [b F, n l] mask
    assert trace(((V // n) * F, n * 64), V, n, F, heads, [64]) == [(64, ((V // n) * F * heads, n * 64, n * 64), ((V // n) * F * heads, n * 64, n * 64), True)]
