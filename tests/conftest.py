import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

# the process-level HIP runtime setting of the product (hardware queues), made the way an application makes it: one
# explicit call before anything touches the GPU
from skghoi_amd import runtime as _runtime  # noqa: E402
_runtime.configure()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir(os.environ.get("SKG_REFERENCE_ROOT", "/root/reference"))
    skip_ref = pytest.mark.skip(reason="reference tree not present (GPU box)")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)


@pytest.fixture(params=["fp32", "fp16x2"])
def precision(request):
    """Inference GEMM path of the heads gpu_run builds: exact fp32 MFMA, or the fp16x2 split-operand loop."""
    import gpu_run
    old = gpu_run.PRECISION
    gpu_run.PRECISION = request.param
    yield request.param
    gpu_run.PRECISION = old
