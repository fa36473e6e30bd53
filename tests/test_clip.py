"""The condition towers (first half): ``animate3d_amd.clip`` against transformers' own CLIP implementation
— the third-party code the reference calls (pipeline.py:345-538, utils/util.py:268-287; the reference pins transformers 4.25.1,
this image has a newer release with the same architecture).  CPU: host logic on the plain-torch op set at small widths, key-for-
key state-dict loading (both key spellings).  GPU: the real SD1.5 text tower (12 x 768, causal, head dim 64) and the real
ViT-H/14 image tower (32 x 1280, head dim 80) on the HIP kernels, bf16 and fp16 storage, random-init weights."""
import pytest
import torch

from animate3d_amd.clip import CLIPTextEncoder, CLIPTowerConfig, CLIPVisionEncoderWithProjection, encode_image, encode_prompt
from tests.torch_ops import TorchRefOps

transformers = pytest.importorskip("transformers")

SMALL_T = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, vocab_size=100, max_position_embeddings=20)
SMALL_V = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=28, patch_size=7, projection_dim=32,
               hidden_act="gelu")


def _hf_text(**kw):
    cfg = transformers.CLIPTextConfig(hidden_act="quick_gelu", bos_token_id=0, eos_token_id=kw.get("vocab_size", 49408) - 1, **kw)
    return transformers.CLIPTextModel(cfg).eval()


def _hf_vision(**kw):
    return transformers.CLIPVisionModelWithProjection(transformers.CLIPVisionConfig(**kw)).eval()


def test_text_tower_host_logic_matches_transformers():
    torch.manual_seed(0)
    hf = _hf_text(**SMALL_T)
    m = CLIPTextEncoder(CLIPTowerConfig(**SMALL_T), ops=TorchRefOps())
    missing, unexpected = m.load_state_dict(hf.state_dict(), strict=True)
    assert not missing and not unexpected
    # the reference's checkpoints (transformers 4.25.1) spell the keys with the "text_model." prefix: both load
    old_style = {("text_model." + k if not k.startswith("text_model.") else k): v for k, v in hf.state_dict().items()}
    m.load_state_dict(old_style, strict=True)
    ids = torch.randint(0, 100, (3, 20))
    with torch.no_grad():
        want = hf(ids)[0]
    got, neg = encode_prompt(m, ids, ids.flip(0))
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(neg, want.flip(0), rtol=1e-4, atol=1e-5)
    # causal: a token's state does not depend on later tokens
    ids2 = ids.clone(); ids2[:, 10:] = 7
    torch.testing.assert_close(m(ids2)[0][:, :10], got[:, :10], rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 21, dtype=torch.long))


def test_vision_tower_host_logic_matches_transformers():
    torch.manual_seed(1)
    hv = _hf_vision(**SMALL_V)
    m = CLIPVisionEncoderWithProjection(CLIPTowerConfig(**SMALL_V), ops=TorchRefOps())
    missing, unexpected = m.load_state_dict(hv.state_dict(), strict=True)
    assert not missing and not unexpected
    px = torch.randn(2, 3, 28, 28)
    with torch.no_grad():
        want = hv(px)
    got = m(px)
    torch.testing.assert_close(got.image_embeds, want.image_embeds, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got.last_hidden_state, want.last_hidden_state, rtol=1e-4, atol=1e-5)
    e, u = encode_image(m, px)
    assert torch.equal(e, got.image_embeds) and float(u.abs().max()) == 0.0 and u.shape == e.shape        # pipeline.py:536-537
    with pytest.raises(ValueError):
        m(torch.randn(1, 3, 35, 28))


def test_real_tower_shapes_and_keys():
    """SD1.5 text encoder: 196 tensors under text_model.*; ViT-H/14: 32 layers, 257 positions, 1280 -> 1024 projection."""
    with torch.device("meta"):
        t, v = CLIPTextEncoder(), CLIPVisionEncoderWithProjection()
    tk, vk = t.state_dict(), v.state_dict()
    assert len(tk) == 2 + 12 * 16 + 2 and tk["text_model.embeddings.token_embedding.weight"].shape == (49408, 768)
    assert tk["text_model.embeddings.position_embedding.weight"].shape == (77, 768)
    assert len(vk) == 3 + 2 + 32 * 16 + 2 + 1 and vk["vision_model.embeddings.position_embedding.weight"].shape == (257, 1280)
    assert vk["visual_projection.weight"].shape == (1024, 1280) and vk["vision_model.embeddings.patch_embedding.weight"].shape == (1280, 3, 14, 14)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,bar", [(torch.bfloat16, 3e-2), (torch.float16, 6e-3)])
def test_text_tower_gpu_parity(dtype, bar):
    torch.manual_seed(0)
    hf = _hf_text(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, vocab_size=49408, max_position_embeddings=77)
    m = CLIPTextEncoder(device="cuda")
    m.load_state_dict(hf.state_dict(), strict=True)
    m = m.to(dtype).eval()
    ids = torch.randint(0, 49408, (4, 77), generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = hf(ids)[0]
    got = m(ids.cuda())[0]
    rel = ((got.float().cpu() - want).norm() / want.norm()).item()
    print(f"[parity] CLIP text tower (12 x 768, causal, D=64) {dtype}: rel_l2={rel:.3e}")
    assert got.shape == (4, 77, 768) and got.dtype == dtype and torch.isfinite(got).all() and rel <= bar


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,bar", [(torch.bfloat16, 3e-2), (torch.float16, 6e-3)])
def test_vision_tower_gpu_parity(dtype, bar):
    torch.manual_seed(1)
    hv = _hf_vision(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224, patch_size=14,
                    projection_dim=1024, hidden_act="gelu")
    m = CLIPVisionEncoderWithProjection(device="cuda")
    m.load_state_dict(hv.state_dict(), strict=True)
    m = m.to(dtype).eval()
    px = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        want = hv(px).image_embeds
    got, unc = encode_image(m, px.cuda())
    rel = ((got.float().cpu() - want).norm() / want.norm()).item()
    print(f"[parity] CLIP ViT-H/14 image tower (32 x 1280, D=80, 257 tokens) {dtype}: image_embeds rel_l2={rel:.3e}")
    assert got.shape == (2, 1024) and torch.isfinite(got).all() and float(unc.abs().max()) == 0.0 and rel <= bar
