import os
import sys

# gradient buffers the kernels must overwrite slice by slice are NaN-filled under test: an unwritten slice fails
os.environ.setdefault("ADP_DEBUG_POISON", "1")

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def _use_emulator():
    from audio_diffusion_pytorch_amd import _C
    sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
    import build_emul
    _C._testing_use_library(build_emul.build(), allow_cpu=True)
    return torch.device("cpu")


def _use_hip():
    from audio_diffusion_pytorch_amd import _C
    _C._testing_use_library(None, allow_cpu=False)
    assert torch.cuda.is_available(), "gpu-marked test without a GPU"
    _C.lib()  # fail loudly if libadp_hip.so is missing
    return torch.device("cuda:0")


@pytest.fixture(params=["emul", pytest.param("hip", marks=pytest.mark.gpu)])
def dev(request):
    """Runs a test twice: against the SIMT-emulated build of the kernel sources on the CPU (logic check,
    not-gpu suite) and against the real gfx950 library on cuda:0 (-m gpu)."""
    from audio_diffusion_pytorch_amd import _C
    d = _use_emulator() if request.param == "emul" else _use_hip()
    yield d
    _C._testing_use_library(None, allow_cpu=False)


@pytest.fixture
def hip():
    from audio_diffusion_pytorch_amd import _C
    d = _use_hip()
    yield d
    _C._testing_use_library(None, allow_cpu=False)


@pytest.fixture
def emul():
    from audio_diffusion_pytorch_amd import _C
    d = _use_emulator()
    yield d
    _C._testing_use_library(None, allow_cpu=False)


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """SURVEY 8d parity metric: ||a-b||_inf / ||b||_inf per tensor."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    den = b.abs().max().item()
    return (a - b).abs().max().item() / (den if den > 0 else 1.0)
