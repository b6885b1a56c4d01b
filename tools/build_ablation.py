"""Build the MEASUREMENT copy of the kernel library: tools/_build/libvqa_hip_ablation.so, compiled with -DVQA_ABLATION so that the
VQA_* environment switches (tile choices, split targets, and the *_DBG phase switches that produce WRONG results on purpose) are
live.  The product library (visual-question-answering-vqa-system_amd/libvqa_hip.so) is compiled without the flag and ignores the
environment.  Usage from a tool:

    import tools.build_ablation as A; A.use()      # before the first kernel call: builds if stale, points _lib at the copy
"""
import importlib
import importlib.util
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "visual-question-answering-vqa-system_amd")
OUT = os.path.join(REPO, "tools", "_build")
LIB = os.path.join(OUT, "libvqa_hip_ablation.so")


def build(force=False):
    spec = importlib.util.spec_from_file_location("vqa_hip_build", os.path.join(PKG, "build.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    srcs = [os.path.join(B.CSRC, f) for f in B.SOURCES]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(s) < os.path.getmtime(LIB) for s in srcs):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    objs, procs = [], []
    for src in srcs:
        obj = os.path.join(OUT, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen([B.HIPCC] + B.FLAGS + ["-DVQA_ABLATION", "-c", src, "-o", obj]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def use():
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    L = importlib.import_module("visual-question-answering-vqa-system_amd._lib")
    L.LIB_PATH = build()
    L._lib = None
    return L


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
