import os
import sys

import pytest

# No bytecode caches from the test runs: the CPU suite imports tests/host_paths.py (the host-side ASAN + UBSan harness, CPU container only), and a
# stale tests/__pycache__/host_paths.*.pyc left behind by a CPU run is the same harness in another encoding — the GPU pool refuses sanitizer
# builds, so it must not travel there (see .gpurunignore). Set before any test module is imported.
sys.dont_write_bytecode = True

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # the parity tests flip EGX_FFN_CUT / EGX_FFN_SLICES / EGX_SLICE_DROP inside one process: the library (which reads them once) re-reads them per call
    from egot2_amd import functional as _F
    _F.reload_tuning_each_call = True
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (this container only)")


@pytest.fixture(scope="session")
def egx_lib():
    """Built + loaded libegot2x.so (build is a no-op when the .so is current)."""
    from egot2_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.load()


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
