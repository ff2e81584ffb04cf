"""Builds libadp_hip.so (the gfx950 kernels + C-ABI of include/adp.h) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the repository snapshot.  No CUDA path, no hipify, no fallback.
"""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libadp_hip.so")
SOURCES = ["conv1d.hip", "conv_mm.hip", "conv_mm_m64.hip", "conv_mm_m32.hip", "conv_tile.hip", "conv_tilek.hip", "conv_tilek1.hip", "conv_mm4.hip", "wgrad_mm.hip", "conv_direct.hip", "wgrad_direct.hip", "norm.hip", "elementwise.hip", "resample.hip", "linear.hip", "attention.hip", "ctx_bank.hip", "probe.hip"]


def _newest_mtime(paths):
    return max(os.path.getmtime(p) for p in paths)


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sources()
    deps = srcs + [os.path.join(CSRC, "adp_rt.h"), os.path.join(CSRC, "conv_internal.h"), os.path.join(CSRC, "conv_mm_impl.h"), os.path.join(REPO_ROOT, "include", "adp.h")]
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= _newest_mtime(deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(PKG_DIR, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(PKG_DIR, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) >= _newest_mtime([s] + deps[len(srcs):]):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-x", "hip", s, "-o", o,
               "-I", os.path.join(REPO_ROOT, "include"), "-I", CSRC, "-Wno-unused-value"]
        if verbose:
            print("[adp build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print("[adp build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
