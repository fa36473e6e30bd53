"""Build libanimate3d_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake.

    python -m animate3d_amd.build            # incremental
    python -m animate3d_amd.build --force
    python -m animate3d_amd.build --experiment A3D_EXP_CHUNK_MAJOR   # side build lib/exp/libanimate3d_hip_<macro>.so for
                                                                      # tools/microbench.py (A3D_LIB=...); never loaded by the package
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libanimate3d_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
EXTRA_FLAGS = {}      # per-file additions, e.g. {"flash_attn.hip": ["-fno-honor-nans"]} (measured: no effect, not used)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path: str) -> str:
    h = hashlib.sha256()
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    for dep in [path, *headers, os.path.join(os.path.dirname(PKG), "include", "animate3d_hip.h")]:
        with open(dep, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + EXTRA_FLAGS.get(os.path.basename(path), [])).encode())
    return h.hexdigest()


# every kernel source is compiled twice: bf16 storage (as is) and IEEE fp16 storage (-DA3D_STORAGE_F16: the a3d_*_f16 entry points)
STORAGE_VARIANTS = (("", []), ("_f16", ["-DA3D_STORAGE_F16"]))


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    objs, jobs = [], []
    for src in sources():
        for suffix, defs in STORAGE_VARIANTS:
            obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + suffix + ".o")
            stamp = obj + ".sha"
            dig = _digest(src) + suffix
            objs.append(obj)
            fresh = (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig)
            if not fresh:
                jobs.append((src, obj, stamp, dig, defs))

    def compile_one(job):
        src, obj, stamp, dig, defs = job
        cmd = [hipcc, *FLAGS, *defs, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        if verbose:
            print("[a3d build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(stamp, "w") as f:
            f.write(dig)

    with ThreadPoolExecutor(max_workers=min(10, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    if jobs or force or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print("[a3d build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


def build_experiment(macro: str, verbose: bool = True) -> str:
    """Compile every source with -D<macro> into lib/exp/ (measurement builds of code that is compiled out of the product)."""
    expdir = os.path.join(LIBDIR, "exp", macro)
    os.makedirs(expdir, exist_ok=True)
    hipcc = _hipcc()
    objs, cmds = [], []
    for src in sources():
        for suffix, defs in STORAGE_VARIANTS:
            obj = os.path.join(expdir, os.path.basename(src)[:-4] + suffix + ".o")
            cmds.append([hipcc, *FLAGS, *defs, "-D" + macro, "-c", src, "-o", obj])
            objs.append(obj)

    def run(cmd):
        if verbose:
            print("[a3d build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=10) as ex:
        list(ex.map(run, cmds))
    lib = os.path.join(LIBDIR, "exp", f"libanimate3d_hip_{macro}.so")
    subprocess.run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", lib], check=True)
    return lib


if __name__ == "__main__":
    if "--experiment" in sys.argv:
        print(build_experiment(sys.argv[sys.argv.index("--experiment") + 1]))
    else:
        print(build(force="--force" in sys.argv))
