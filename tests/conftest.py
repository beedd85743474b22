import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=["f32", "bf16x3"])
def gemm_mode(request):
    """Run a GPU test under both arithmetic modes of the dense layers (include/b4r.h b4r_set_gemm_mode)."""
    from bert4rec_amd import _lib
    lib = _lib.load()
    prev = lib.b4r_get_gemm_mode()
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_F32 if request.param == "f32" else _lib.GEMM_BF16X3))
    yield request.param
    lib.b4r_set_gemm_mode(prev)
