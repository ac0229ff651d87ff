import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# torch ships its own HIP runtime (libamdhip64.so.7, ROCm 7.0) and libcamsync_hip.so links the system one (ROCm 7.2)
# under the same soname: whichever is loaded first serves both.  Tests that use torch for rendering or
# torch.distributed need torch's; loading torch before any test can load the library keeps that independent of the
# order and selection of the tests (a run that loaded the library first made torch report "No HIP GPUs").
import torch  # noqa: E402,F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
