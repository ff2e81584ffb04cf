"""ctypes binding of libadp_hip.so (C-ABI in include/adp.h).

The product path has exactly one backend: the hand-written gfx950 kernels.  If the shared
library is missing or a tensor is not on a HIP device, calls fail loudly -- there is no CPU
or PyTorch fallback.  (`_testing_use_library` exists so the GPU-less unit tests can point the
same host code at the SIMT-emulated build of the *same kernel sources*; see tests/emul/.)
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int, c_int64, c_void_p

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libadp_hip.so")

_lib = None
_allow_cpu = False

ERRORS = {-1: "bad shape", -2: "unsupported configuration", -3: "misaligned pointer", -4: "kernel launch failed",
          -5: "null pointer"}

P = c_void_p
I = c_int64
F = c_float


class ConvDesc(Structure):
    _fields_ = [(n, c_void_p) for n in ("x", "x2", "w", "bias", "pro_stats", "pro_gamma", "pro_beta", "e_scale",
                                        "res", "out", "out_pre")] + \
               [(n, c_int64) for n in ("B", "R", "R1", "Lin", "M", "N", "KT", "stride", "dil", "pad", "up",
                                       "transposed", "prologue", "groups", "store", "sp", "e_bstride")] + \
               [("ws", c_void_p), ("gn_part", c_void_p)] + \
               [(n, c_void_p) for n in ("gnb_x", "gnb_stats", "gnb_gamma", "gnb_beta", "gnb_ab")] + [("gnb_groups", c_int64)]


class WgradDesc(Structure):
    _fields_ = [(n, c_void_p) for n in ("x", "x2", "dy", "pro_stats", "pro_gamma", "pro_beta", "dw", "dbias", "ws")] + \
               [(n, c_int64) for n in ("B", "R", "R1", "Lin", "M", "N", "KT", "stride", "dil", "pad", "up",
                                       "prologue", "groups", "accumulate")]


# name -> (restype, argtypes); mirrors include/adp.h one to one
SIGNATURES = {
    "adp_version": (c_int, []),
    "adp_launch_trace": (I, [I, ctypes.c_char_p, I]),
    "adp_launch_times": (I, [P, I]),
    "adp_conv1d_ws_bytes": (I, [POINTER(ConvDesc)]),
    "adp_conv1d_gn_entries": (I, [POINTER(ConvDesc)]),
    "adp_conv1d_gnb_entries": (I, [POINTER(ConvDesc)]),
    "adp_conv1d": (c_int, [POINTER(ConvDesc), P]),
    "adp_conv1d_tile": (I, [POINTER(ConvDesc)]),
    "adp_conv1d_wgrad_ws_bytes": (I, [POINTER(WgradDesc)]),
    "adp_conv1d_wgrad": (c_int, [POINTER(WgradDesc), P]),
    "adp_conv1d_wgrad_batch": (c_int, [POINTER(WgradDesc), I, P]),
    "adp_conv1d_wgrad_partials": (I, [POINTER(WgradDesc)]),
    "adp_wgrad_reduce_batch": (c_int, [P, P, P, I, I, I, I, I, P]),
    "adp_gn_stats_ws_bytes": (I, [I, I, I, I]),
    "adp_gn_stats": (c_int, [P, I, I, I, I, F, P, P, P]),
    "adp_gn_stats_act": (c_int, [P, I, I, I, I, F, P, P, P, P, P, P]),
    "adp_gn_finalize": (c_int, [P, I, I, I, I, F, P, P]),
    "adp_gn_finalize_act": (c_int, [P, P, I, I, I, I, I, F, P, P, P, P, P]),
    "adp_gn_act": (c_int, [P, P, P, P, I, I, I, I, P, P]),
    "adp_row_nsplit": (I, [I, I]),
    "adp_gn_silu_bwd_reduce": (c_int, [P, P, P, P, P, I, I, I, I, I, P, P]),
    "adp_gn_silu_bwd_apply": (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, I, P]),
    "adp_gn_silu_bwd_apply_ab": (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, I, P, P, P, I, P]),
    "adp_gn_param_grad": (c_int, [P, I, I, I, P, P, I, P]),
    "adp_modulation_fwd": (c_int, [P, P, I, I, I, I, F, P, P, P]),
    "adp_modulation_ln_fwd": (c_int, [P, P, I, I, I, I, F, P, P, F, P, P, P, P, P, P, P, P]),
    "adp_chan_ln_bwd_ws_bytes": (I, [I, I, I]),
    "adp_modulation_bwd": (c_int, [P, P, P, I, P, I, I, I, P, P, I, P, P]),
    "adp_modulation_bwd_partial": (I, [P, P, P, I, P, I, I, I, P, P, P]),
    "adp_modulation_bwd_reduce": (c_int, [P, P, I, I, I, I, I, P]),
    "adp_modulation_ln_bwd_partial": (I, [P, P, I, P, P, P, P, P, P, I, I, I, I, P, P, P, P, P]),
    "adp_ln_stats": (c_int, [P, I, I, I, F, P, P]),
    "adp_ln_affine_fwd": (c_int, [P, I, I, I, F, P, P, P, P, P, P, P, P]),
    "adp_ln_bwd": (c_int, [P, P, P, P, P, I, I, I, I, P, P, P, P]),
    "adp_linear_fwd": (c_int, [P, P, P, I, I, I, I, I, P, I, P]),
    "adp_linear_bwd_data_ws_bytes": (I, [I, I, I]),
    "adp_linear_bwd_data": (c_int, [P, I, P, I, I, I, I, P, P, P]),
    "adp_linear_bwd_weight": (c_int, [P, I, P, I, I, I, I, I, P, P, P]),
    "adp_time_fourier_fwd": (c_int, [P, P, I, I, P, P]),
    "adp_time_fourier_bwd": (c_int, [P, P, P, I, I, I, P, P]),
    "adp_act_fwd": (c_int, [P, I, I, P, P]),
    "adp_act_bwd": (c_int, [P, P, I, I, I, P, P]),
    "adp_skipmod_bwd_ws_bytes": (I, [I, I, I]),
    "adp_skipmod_bwd": (c_int, [P, P, P, I, I, I, I, P, P, I, P, P]),
    "adp_v_noise": (c_int, [P, P, P, I, I, P, P, P]),
    "adp_mse_ws_bytes": (I, [I]),
    "adp_mse_fwd": (c_int, [P, P, I, P, P, P]),
    "adp_mse_bwd": (c_int, [P, P, P, I, P, P]),
    "adp_v_step": (c_int, [P, P, P, I, P, P]),
    "adp_add": (c_int, [P, P, I, P, P]),
    "adp_axpby": (c_int, [F, P, F, P, I, P, P]),
    "adp_copy2d": (c_int, [P, I, P, I, I, I, P]),
    "adp_unshuffle": (c_int, [P, I, I, I, P, P]),
    "adp_pool_sum": (c_int, [P, I, I, I, P, P, P]),
    "adp_v_inpaint_step": (c_int, [P, P, P, P, P, P, I, P, P]),
    "adp_cfg_mix": (c_int, [P, I, F, P, P]),
    "adp_select_rows": (c_int, [P, P, P, I, I, P, P]),
    "adp_resample": (c_int, [P, P, I, I, I, I, I, I, I, P, P]),
    "adp_ctx_fold_fwd": (c_int, [P, P, P, I, I, I, P, P, P]),
    "adp_ctx_fold_bwd": (c_int, [P, P, P, P, P, I, I, I, P, P, P, P]),
    "adp_attn_fwd": (c_int, [P, P, P, I, I, I, I, I, I, I, P, P, P, P]),
    "adp_attn_fwd_ws_bytes": (I, [I, I, I, I, I]),
    "adp_attn_bwd_ws_bytes": (I, [I, I, I, I, I]),
    "adp_attn_bwd": (c_int, [P, P, P, P, P, P, I, I, I, I, I, I, I, P, P, P, P, P]),
    "adp_probe_copy": (c_int, [P, P, I, P]),
    "adp_probe_mfma": (I, [I, P, I, P]),
    "adp_probe_launch": (c_int, [I, P]),
    "adp_probe_chase": (c_int, [P, I, P, P]),
    "adp_probe_copy_v": (c_int, [P, P, I, c_int, P]),
    "adp_probe_mfma_v": (I, [I, P, I, c_int, P]),
}


def _bind(path: str):
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export what adp.h declares
        fn.restype = res
        fn.argtypes = args
    return lib


def lib():
    """The loaded kernel library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the gfx950 kernels first "
                "(python -m audio_diffusion_pytorch_amd.build or __graft_entry__.build()); "
                "audio_diffusion_pytorch_amd has no CPU/PyTorch fallback path")
        _lib = _bind(LIB_PATH)
    return _lib


def _testing_use_library(path, allow_cpu: bool):
    """TESTS ONLY: route the host code to another build of the same C-ABI (the SIMT emulator)."""
    global _lib, _allow_cpu
    _lib = _bind(path) if path is not None else None
    _allow_cpu = allow_cpu


def stream() -> int:
    if _allow_cpu and not torch.cuda.is_available():
        return 0
    return torch.cuda.current_stream().cuda_stream


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous fp32 (or `dtype`) tensor (None -> NULL)."""
    if t is None:
        return None
    if t.dtype != dtype:
        raise TypeError(f"adp kernel argument must be {dtype}; got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("adp kernels need contiguous tensors")
    if not t.is_cuda and not _allow_cpu:
        raise RuntimeError("adp kernels run on MI355X (HIP) tensors only; got a CPU tensor and there is no CPU path")
    return t.data_ptr()


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what} failed: {ERRORS.get(code, code)} ({code})")


# ---- launch profiling (bench.py's roofline leg): when PROFILE is a list, every C-ABI call is bracketed by HIP
# events on the stream it launches on and labelled with the kernel instantiation(s) it dispatched.
PROFILE = None
_TAG = None
# bench.py's convblock leg: when REPLAY is a list, ops.conv1d appends (shape label, algorithmic bytes, relaunch callable) for
# every conv it issues, so that single launches can be re-timed from a hipGraph without the per-launch event pair
REPLAY = None


def tag(**meta):
    """Attach algorithmic flops / bytes / a shape label to the next profiled call (no-op when not profiling)."""
    global _TAG
    if PROFILE is not None:
        _TAG = meta


def _decode_trace(text: str) -> str:
    """'(kern<BM, BN>)@int ns::launch(...) [BM = 64, BN = 32]' -> 'kern<64, 32>' (rocprofv3's spelling)."""
    import re
    names = []
    for item in text.split("\n"):
        if not item:
            continue
        kern, _, site = item.partition("@")
        kern = kern.strip()
        while kern.startswith("(") and kern.endswith(")"):
            kern = kern[1:-1].strip()
        m = re.search(r"\[(?:with )?([^\]]*)\]\s*$", site)
        env = {}
        if m:
            for kv in m.group(1).replace(";", ",").split(","):
                k, _, v = kv.partition("=")
                if v:
                    env[k.strip().split()[-1]] = v.strip()
        if "<" in kern:
            head, _, rest = kern.partition("<")
            args = [a.strip() for a in rest.rsplit(">", 1)[0].split(",")]
            kern = head + "<" + ", ".join(env.get(a, a) for a in args) + ">"
        names.append(kern)
    return " + ".join(names)


def call(name: str, *args):
    global _TAG
    if PROFILE is None:
        check(getattr(lib(), name)(*args), name)
        return
    l = lib()
    l.adp_launch_trace(1, None, 0)      # recording on: names + a HIP event pair per launch, on the launch stream
    check(getattr(l, name)(*args), name)
    buf = ctypes.create_string_buffer(4096)
    l.adp_launch_trace(0, buf, 4096)
    kernels = [k for k in _decode_trace(buf.value.decode()).split(" + ") if k]
    PROFILE.append([name, kernels, _TAG or {}, None])   # the last slot receives the per-kernel times
    _TAG = None


def call_value(name: str, *args) -> int:
    """`call` for entry points that launch AND return a non-negative value (negative: error code)."""
    global _TAG
    l = lib()
    if PROFILE is not None:
        l.adp_launch_trace(1, None, 0)
    v = getattr(l, name)(*args)
    if PROFILE is not None:
        buf = ctypes.create_string_buffer(4096)
        l.adp_launch_trace(0, buf, 4096)
        kernels = [k for k in _decode_trace(buf.value.decode()).split(" + ") if k]
        PROFILE.append([name, kernels, _TAG or {}, None])
        _TAG = None
    if v < 0:
        raise RuntimeError(f"{name} failed: {ERRORS.get(v, v)} ({v})")
    return int(v)


def profile_collect():
    """Ends a profiled region: waits for the recorded events and returns [(call, kernel, meta, ms), ...] with one
    entry per kernel launch (meta -- algorithmic flops / bytes / shape -- is attached to a call's first kernel)."""
    global PROFILE
    recs, PROFILE = PROFILE or [], None
    n = sum(len(r[1]) for r in recs)
    arr = (c_float * max(n, 1))()
    got = lib().adp_launch_times(arr, n)
    out, i = [], 0
    for name, kernels, meta, _ in recs:
        for j, k in enumerate(kernels):
            out.append((name, k, meta if j == 0 else {}, arr[i] if i < got else float("nan")))
            i += 1
    return out


def query(name: str, *args) -> int:
    v = getattr(lib(), name)(*args)
    if v < 0:
        raise RuntimeError(f"{name} failed: {ERRORS.get(v, v)} ({v})")
    return int(v)
