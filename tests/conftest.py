import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import ora
    ora.build()
    return ora


@pytest.fixture(scope="session")
def gpu():
    """Initialise the HIP module on cuda:0; fails loudly if the extension is missing."""
    import torch
    from llamafile_amd import _hip, sgemm
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    _hip.lib()  # raises if libllamafile_amd_hip.so is not built
    sgemm.init(0)
    return sgemm
