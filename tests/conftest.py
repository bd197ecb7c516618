import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _release_device_buffers():
    yield
    try:
        import gpu_util
        if gpu_util._KEEP:
            gpu_util.release()
    except Exception:
        pass
