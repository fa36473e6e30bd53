"""Generate tests/golden/processors.npz by running the REFERENCE's own attention processors.

Run in the build container only (needs /root/reference):

    python -B tests/golden/make_processor_goldens.py

The reference's ``animatediff/models/attention_processor.py`` and ``embeddings.py`` are
imported from /root/reference unmodified.  The third-party names they import that are not
installed here (diffusers 0.28.0, xformers 0.0.16) are replaced IN MEMORY by small
stand-ins: diffusers' ``Attention`` container / ``AlphaBlender`` /
``SinusoidalPositionalEmbedding`` restated in ``oracle/unet_ref.py`` and
``memory_efficient_attention := scaled_dot_product_attention``.  What the vectors pin is
therefore the reference's processor logic (token regrouping, first-frame K/V selection, PE
placement, output-projection order, blend direction), not diffusers itself.

Only data (seeded inputs, weights and the reference's outputs) is written; no reference
source leaves /root/reference.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import unet_ref as O  # noqa: E402
from tests.golden.seeded import fill_named, seeded_tensor  # noqa: E402

REF = "/root/reference"


def _install_standins():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    d = mod("diffusers")
    du = mod("diffusers.utils")
    du.USE_PEFT_BACKEND = False
    dm = mod("diffusers.models")
    dap = mod("diffusers.models.attention_processor")
    dap.Attention = O.Attention
    de = mod("diffusers.models.embeddings")

    class SinusoidalPositionalEmbedding(O.TimePosEmbed):
        def __init__(self, embed_dim, max_seq_length=32):
            super().__init__(embed_dim, max_seq_length)

    class LabelEmbedding(nn.Module):          # only constructed when camera encoding is on (off in release)
        def __init__(self, num_classes, hidden_size, dropout_prob):
            super().__init__()
            self.embedding_table = nn.Embedding(num_classes + int(dropout_prob > 0), hidden_size)

        def forward(self, labels):
            return self.embedding_table(labels)

    de.SinusoidalPositionalEmbedding = SinusoidalPositionalEmbedding
    de.LabelEmbedding = LabelEmbedding
    dr = mod("diffusers.models.resnet")

    class AlphaBlender(O.AlphaBlender):
        def __init__(self, alpha, merge_strategy="learned", switch_spatial_to_temporal_mix=False):
            assert merge_strategy == "learned" and not switch_spatial_to_temporal_mix
            super().__init__(alpha)

        def forward(self, x_spatial, x_temporal, image_only_indicator=None):
            return super().forward(x_spatial, x_temporal)

    dr.AlphaBlender = AlphaBlender
    d.utils, d.models = du, dm
    dm.attention_processor, dm.embeddings, dm.resnet = dap, de, dr
    xf = mod("xformers")
    xo = mod("xformers.ops")

    def memory_efficient_attention(q, k, v, attn_bias=None, op=None, scale=None):
        return F.scaled_dot_product_attention(q, k, v, attn_mask=attn_bias, scale=scale)

    xo.memory_efficient_attention = memory_efficient_attention
    xf.ops = xo


def _rand(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


def _fill(module, gen):
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("mix_factor"):
                p.copy_(torch.tensor([0.3]))
            else:
                p.copy_(_rand(gen, *p.shape, scale=0.2))


def main():
    _install_standins()
    sys.path.insert(0, REF)
    from animatediff.models import attention_processor as RP     # the reference's code
    from animatediff.models.embeddings import SinePositionalEncoding2D

    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(1234)
    b, n, f, fs, C, H, CROSS = 1, 2, 4, 4, 32, 4, 24
    L = fs * fs
    out = {"meta": np.array([b, n, f, fs, C, H, CROSS], dtype=np.int64)}

    def save_sd(prefix, module):
        for k, v in module.state_dict().items():
            out[f"{prefix}/{k}"] = v.detach().numpy().copy()

    with torch.no_grad():
        # ---- a10: SinePositionalEncoding2D(normalize=True) on a non-square map too
        for (hh, ww) in [(fs, fs), (3, 5)]:
            pe = SinePositionalEncoding2D(C // 2, normalize=True)._forward(torch.zeros(1, hh, ww))
            out[f"sine2d/{hh}x{ww}"] = pe[0].numpy().copy()

        # ---- a7: MVDreamXFormersAttnProcessor
        attn = O.Attention(C, None, H, C // H)
        _fill(attn, gen)
        x = _rand(gen, b * n * f, L, C)
        proc = RP.MVDreamXFormersAttnProcessor(num_views=n, num_frames=f)
        out["mvdream/x"] = x.numpy().copy()
        out["mvdream/y"] = proc(attn, x).numpy().copy()
        save_sd("mvdream/attn", attn)

        # ---- a6: MVDreamI2VXFormersAttnProcessor
        attn = O.Attention(C, None, H, C // H)
        _fill(attn, gen)
        proc = RP.MVDreamI2VXFormersAttnProcessor(hidden_size=C, num_views=n, num_frames=f)
        _fill(proc, gen)
        x = _rand(gen, b * n * f, L, C)
        out["mvi2v/x"] = x.numpy().copy()
        out["mvi2v/y"] = proc(attn, x).numpy().copy()
        save_sd("mvi2v/attn", attn)
        save_sd("mvi2v/proc", proc)

        # ---- a8: IPAdapterXFormersAttnProcessor
        attn = O.Attention(C, CROSS, H, C // H)
        _fill(attn, gen)
        proc = RP.IPAdapterXFormersAttnProcessor(hidden_size=C, cross_attention_dim=CROSS, num_tokens=(4,), scale=0.7)
        _fill(proc, gen)
        x = _rand(gen, b * n * f, L, C)
        text = _rand(gen, b * n * f, 7, CROSS)
        ip = _rand(gen, b * n * f, 4, CROSS)
        out["ipadapter/x"], out["ipadapter/text"], out["ipadapter/ip"] = x.numpy().copy(), text.numpy().copy(), ip.numpy().copy()
        out["ipadapter/y"] = proc(attn, x, encoder_hidden_states=(text, [ip])).numpy().copy()
        save_sd("ipadapter/attn", attn)
        save_sd("ipadapter/proc", proc)

        # ---- a9: SpatioTemporalI2VXFormersAttnProcessor, released switches (+ no-blender variant)
        NS = types.SimpleNamespace
        for tag, blender in (("st_blend", True), ("st_sum", False)):
            spatial_cfg = NS(enabled=True, attn_cfg=NS(use_spatial_encoding=True, spatial_encoding_type="sinusoid",
                                                        use_camera_encoding=False, camera_encoding_type="sinusoid"))
            image_cfg = NS(enabled=False)
            attn = O.Attention(C, None, H, C // H)
            _fill(attn, gen)
            proc = RP.SpatioTemporalI2VXFormersAttnProcessor(hidden_size=C, feature_size=fs, num_views=n, num_frames=f,
                                                             spatial_attn=spatial_cfg, image_attn=image_cfg,
                                                             use_alpha_blender=blender)
            _fill(proc, gen)
            x = _rand(gen, b * n * L, f, C)
            out[f"{tag}/x"] = x.numpy().copy()
            out[f"{tag}/y"] = proc(attn, x).numpy().copy()
            save_sd(f"{tag}/attn", attn)
            save_sd(f"{tag}/proc", proc)

        # ---- a9, switch sets the released configs leave off (name-seeded weights, tests/golden/seeded.py): first-frame image
        #      branch with the 3-way SoftmaxAlphaBlender / 2-way blend / plain sum, camera encodings, learnable 2-D encoding,
        #      spatial attention without any encoding.  The learnable camera branch calls .cuda() (attention_processor.py:568):
        #      made a no-op here.
        def st_case(tag, C_, H_, n_, f_, fs_, blender, spatial=True, image=False, enc=True, enc_type="sinusoid", cam=False,
                    cam_type="sinusoid", scale=0.2):
            spatial_cfg = NS(enabled=spatial, attn_cfg=NS(use_spatial_encoding=enc, spatial_encoding_type=enc_type,
                                                          use_camera_encoding=cam, camera_encoding_type=cam_type))
            attn = fill_named(O.Attention(C_, None, H_, C_ // H_), f"{tag}/attn", scale)
            proc = RP.SpatioTemporalI2VXFormersAttnProcessor(hidden_size=C_, feature_size=fs_, num_views=n_, num_frames=f_,
                                                             spatial_attn=spatial_cfg, image_attn=NS(enabled=image),
                                                             use_alpha_blender=blender)
            fill_named(proc, f"{tag}/proc", scale)
            x = seeded_tensor(f"{tag}/x", (b * n_ * fs_ * fs_, f_, C_))
            out[f"{tag}/x"] = x.numpy().copy()
            out[f"{tag}/y"] = proc(attn, x).numpy().copy()

        orig_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            st_case("st_img3", C, H, n, f, fs, True, image=True)
            st_case("st_imgonly", C, H, n, f, fs, True, spatial=False, image=True)
            st_case("st_sum3", C, H, n, f, fs, False, image=True)
            st_case("st_cam_sin", C, H, n, f, fs, True, cam=True, cam_type="sinusoid")
            st_case("st_cam_learn", C, H, n, f, fs, True, cam=True, cam_type="learnable")
            st_case("st_camonly", C, H, n, f, fs, True, enc=False, cam=True, cam_type="learnable")
            st_case("st_learn2d", C, H, n, f, fs, True, enc_type="learnable")
            st_case("st_noenc", C, H, n, f, fs, True, enc=False)
        finally:
            torch.Tensor.cuda = orig_cuda

        # ---- real head dims (D = 40: C = 320, D = 80: C = 640; 8 heads) at a tiny token geometry: vectors the HIP kernels can
        #      replay directly (tests/test_reference_vectors_gpu.py).  Weights are name-seeded, only x / y are stored.
        n2, f2, fs2 = 2, 2, 4
        L2 = fs2 * fs2
        for C_ in (320, 640):
            tag = f"mvi2v_c{C_}"
            attn = fill_named(O.Attention(C_, None, 8, C_ // 8), f"{tag}/attn", C_ ** -0.5)
            proc = fill_named(RP.MVDreamI2VXFormersAttnProcessor(hidden_size=C_, num_views=n2, num_frames=f2), f"{tag}/proc", C_ ** -0.5)
            x = seeded_tensor(f"{tag}/x", (b * n2 * f2, L2, C_))
            out[f"{tag}/x"], out[f"{tag}/y"] = x.numpy().copy(), proc(attn, x).numpy().copy()

            tag = f"ip_c{C_}"
            attn = fill_named(O.Attention(C_, 768, 8, C_ // 8), f"{tag}/attn", 0.04)
            proc = fill_named(RP.IPAdapterXFormersAttnProcessor(hidden_size=C_, cross_attention_dim=768, num_tokens=(4,), scale=0.7),
                              f"{tag}/proc", 0.04)
            x = seeded_tensor(f"{tag}/x", (b * n2 * f2, L2, C_))
            text, ip = seeded_tensor(f"{tag}/text", (b * n2 * f2, 77, 768)), seeded_tensor(f"{tag}/ip", (b * n2 * f2, 4, 768))
            # the product projects text / IP tokens once per VIDEO: make the per-frame copies identical, as the UNet does (:754,763)
            text = text.reshape(b * n2, f2, 77, 768)[:, :1].expand(-1, f2, -1, -1).reshape(b * n2 * f2, 77, 768).contiguous()
            ip = ip.reshape(b * n2, f2, 4, 768)[:, :1].expand(-1, f2, -1, -1).reshape(b * n2 * f2, 4, 768).contiguous()
            out[f"{tag}/x"], out[f"{tag}/text"], out[f"{tag}/ip"] = x.numpy().copy(), text[::f2].numpy().copy(), ip[::f2].numpy().copy()
            out[f"{tag}/y"] = proc(attn, x, encoder_hidden_states=(text, [ip])).numpy().copy()

            st_case(f"st_c{C_}", C_, 8, n2, f2, fs2, True, scale=C_ ** -0.5)
            st_case(f"st_img3_c{C_}", C_, 8, n2, f2, fs2, True, image=True, scale=C_ ** -0.5)
        out["meta_real"] = np.array([b, n2, f2, fs2], dtype=np.int64)

        # ---- embeddings.LearnedPositionalEncoding2D (embeddings.py:99-157) on a non-square map
        from animatediff.models.embeddings import LearnedPositionalEncoding2D
        lp = fill_named(LearnedPositionalEncoding2D(C // 2, row_num_embed=6, col_num_embed=7), "learned2d", 1.0)
        out["learned2d/3x5"] = lp._forward(torch.zeros(1, 3, 5))[0].numpy().copy()

        # ---- pipeline.get_camera (pipeline.py:127-190): torch-only helpers, extracted by exec of
        #      just those three function definitions (the module itself imports diffusers/torchvision).
        import ast, math  # noqa: E401
        src = open(os.path.join(REF, "animatediff/pipelines/pipeline.py")).read()
        tree = ast.parse(src)
        wanted = [nd for nd in tree.body if isinstance(nd, ast.FunctionDef) and nd.name in ("get_camera", "generate_c2w", "normalize_camera")]
        ns = {"torch": torch, "np": np, "math": math, "F": F}
        exec(compile(ast.Module(body=wanted, type_ignores=[]), "pipeline_camera", "exec"), ns)
        for nv in (4, 8):
            out[f"camera/{nv}"] = ns["get_camera"](nv).numpy().copy()

    path = os.path.join(HERE, "processors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if not k.count("attn") and not k.count("proc")})


if __name__ == "__main__":
    main()
