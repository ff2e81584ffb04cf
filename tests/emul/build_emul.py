"""TEST INFRASTRUCTURE: compiles the product kernel sources with g++ -DADP_EMULATE against the
host-side SIMT emulator (adp_rt_emul.h) into tests/emul/libadp_emul.so, so CPU-only tests can
exercise tiling / indexing / reduction logic of the HIP kernels through the same C-ABI.
Never imported by the package; never a fallback for the product path."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO_ROOT, "audio_diffusion_pytorch_amd", "csrc")
LIB_PATH = os.path.join(HERE, "libadp_emul.so")
SOURCES = ["conv1d.hip", "conv_mm.hip", "conv_mm_m64.hip", "conv_mm_m32.hip", "conv_tile.hip", "conv_tilek.hip", "conv_tilek1.hip", "conv_mm4.hip", "wgrad_mm.hip", "conv_direct.hip", "wgrad_direct.hip", "norm.hip", "elementwise.hip", "resample.hip", "linear.hip", "attention.hip", "ctx_bank.hip", "probe.hip"]


def build(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, "adp_rt.h"), os.path.join(CSRC, "conv_internal.h"), os.path.join(CSRC, "conv_mm_impl.h"), os.path.join(HERE, "adp_rt_emul.h"),
                   os.path.join(REPO_ROOT, "include", "adp.h")]
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
        return LIB_PATH
    objs, procs = [], []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-DADP_EMULATE", "-x", "c++", "-c", s, "-o", o,
               "-I", os.path.join(REPO_ROOT, "include"), "-I", CSRC, "-I", HERE, "-Wno-attributes",
               "-Wno-unknown-pragmas", "-ffp-contract=off"]
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("g++ failed: " + " ".join(cmd))
    subprocess.check_call(["g++", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True))
