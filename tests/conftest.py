import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def gpu():
    """The product package on a GPU box.  Fails loudly if the HIP library is missing."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import gpusorting_amd
    return gpusorting_amd


@pytest.fixture(params=["default-routing", "general-path", "position-chains"])
def routing(request, monkeypatch):
    """Three routings of the same test: the library's own (single-tile kernel for small n, the two-launch MSD + bucket sort
    up to 2^20 keys, the general pipeline above); the mid-size route switched off, so that the general pipeline stays covered
    at the sizes the mid-size route takes; and every sort of 2^20 32-bit keys or more forced onto the position-chain plan
    (PF_POS: all passes on position chains, each counting the next one's digit while it scatters — the plan skewed keys get at
    2^25 keys and more), whatever the keys look like.  tests/test_gpu_parity.py runs every test under all three; the large
    cases of the other GPU test files ask for it by name."""
    if request.param == "general-path":
        monkeypatch.setenv("GPUSORT_MID_PATH", "0")
    if request.param == "position-chains":
        monkeypatch.setenv("GPUSORT_MID_PATH", "0")
        monkeypatch.setenv("GPUSORT_POS", "2")
        monkeypatch.setenv("GPUSORT_POS_MIN_LOG2", "20")
    return request.param
