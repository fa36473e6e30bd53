"""CPU checks of the C-ABI boundary: the library builds for gfx950, loads, and exports exactly the
symbols include/animate3d_hip.h declares (no compute calls: there is no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "animate3d_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(a3d_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from animate3d_amd import build
    return build.build(verbose=False)


def test_header_symbols_are_exported(lib_path):
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], check=True, capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (a3d_[a-z0-9_]+)", out)))
    assert exported == _declared(), (exported, _declared())


def test_ctypes_binding_covers_header(lib_path):
    from animate3d_amd import hip_ops
    assert sorted(hip_ops.EXPORTED_SYMBOLS) == _declared()
    lib = hip_ops.load_library()
    assert lib.a3d_version().decode().startswith("animate3d_hip gfx950")
    assert lib.a3d_group_norm_ws_floats(2, 300, 32) == 2 * 3 * 64 + 2 * 64


def test_code_object_is_gfx950(lib_path):
    out = subprocess.run(["strings", "-a", lib_path], capture_output=True, text=True).stdout
    assert "gfx950" in out
    assert not re.search(r"gfx9(0a|42|40)\b", out)


def test_product_has_no_cpu_fallback():
    """On a box without a GPU the product op set must refuse to run rather than fall back."""
    import torch
    from animate3d_amd.hip_ops import HipOps
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        HipOps()


def test_product_never_imports_oracle_or_tests():
    pkg = os.path.join(ROOT, "animate3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", src, flags=re.M), f


def test_geglu_interleave_roundtrip_and_rowmap_reference():
    """Host-side helpers of the boundary that need no GPU: the GEGLU weight interleave is a permutation that the
    reference op set inverts, and the RowMap formula of include/animate3d_hip.h equals the einops regrouping
    "(b n f) l -> (b f) (n l)" it replaces (attention_processor.py:54,340)."""
    import torch
    from animate3d_amd.hip_ops import HipOps, RowMap
    from tests.torch_ops import TorchRefOps, rowmap_indices
    w = torch.arange(128 * 3, dtype=torch.float32).reshape(128, 3)
    il = HipOps.interleave_geglu(w)
    assert torch.equal(TorchRefOps.deinterleave_geglu(il), w)
    assert torch.equal(il[:32], w[:32]) and torch.equal(il[32:64], w[64:96])
    b, n, f, L = 2, 3, 4, 5
    rows = torch.arange(b * n * f * L).reshape(b, n, f, L)
    want = rows.permute(0, 2, 1, 3).reshape(b * f, n * L)                     # "(b n f) l -> (b f) (n l)"
    got = rowmap_indices(RowMap(f, n * f * L, L, L, f * L), b * f, n * L)
    assert torch.equal(got, want)
    first = rows[:, :, 0:1].expand(b, n, f, L).permute(0, 2, 1, 3).reshape(b * f, n * L)   # first-frame K/V, :389-397
    assert torch.equal(rowmap_indices(RowMap(f, n * f * L, 0, L, f * L), b * f, n * L), first)


def _device_kernels(obj_path, workdir):
    """{kernel symbol: [(address, instruction text)]} of the gfx950 code object bundled in a host object file."""
    import shutil
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    local = os.path.join(workdir, os.path.basename(obj_path))
    shutil.copy(obj_path, local)
    subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=workdir)
    bundle = [f for f in os.listdir(workdir) if f.startswith(os.path.basename(obj_path) + ".") and "gfx950" in f]
    assert len(bundle) == 1, bundle
    text = subprocess.run([objdump, "-d", os.path.join(workdir, bundle[0])], check=True, capture_output=True, text=True).stdout
    kernels, cur, base = {}, None, 0
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]{16}) <(\S+)>:", line)
        if m:
            base, cur = int(m.group(1), 16), m.group(2)
            kernels[cur] = {"base": base, "ins": []}
            continue
        m = re.match(r"^\s+(\S.*?)\s*// ([0-9A-F]{12}):", line)
        if m and cur:
            kernels[cur]["ins"].append((int(m.group(2), 16), m.group(1), line))
    return kernels


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs llvm-objdump")
@pytest.mark.parametrize("stem", ["flash_attn_dm", "flash_attn_dm80", "flash_attn_dm160"])
@pytest.mark.parametrize("suffix", ["", "_f16"])
def test_attention_key_loops_do_not_touch_scratch(lib_path, tmp_path, stem, suffix):
    """The LDS-DMA staged attention kernels count their own vmcnt: a register spilled inside the key loop comes back through scratch_load +
    s_waitcnt vmcnt(0), i.e. behind the tile pieces requested just before it — a full memory round trip per iteration (round 6: two spilled
    DMA offsets cost the head_dim-160 kernel 25 %).  Spills in the prologue / epilogue (and in the exact fallback pass of the head_dim-160 kernel)
    are tolerated; the straight-line pipelined loops that hold the max-free pass's MFMAs must have none."""
    obj = os.path.join(os.path.dirname(lib_path), "obj", stem + suffix + ".o")
    kernels = _device_kernels(obj, str(tmp_path))
    checked = 0
    for name, k in kernels.items():
        if "kernel" not in name:
            continue
        ins = k["ins"]
        addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
        for i, (a, txt, line) in enumerate(ins):
            if not txt.startswith(("s_cbranch", "s_branch")):
                continue
            m = re.search(r"<\S+\+0x([0-9a-f]+)>", line)
            if not m:
                continue
            target = k["base"] + int(m.group(1), 16)
            if target >= a or target not in addr_index:
                continue                                   # forward branch
            body = ins[addr_index[target]:i + 1]
            if sum(t.startswith("v_mfma") for _, t, _ in body) < 16 or any(t.startswith(("s_cbranch", "s_branch")) for _, t, _ in body[:-1]):
                continue                                   # not a pipelined key loop (those are straight-line code; the exact fallback pass branches)
            checked += 1
            spills = [t for _, t, _ in body if t.startswith("scratch_")]
            assert not spills, f"{name}: {len(spills)} scratch instructions inside a key loop, e.g. {spills[0]}"
    assert checked >= 1, f"no key loop found in {stem + suffix}"
