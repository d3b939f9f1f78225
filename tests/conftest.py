import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def afv():
    """the product package (directory name has a hyphen)"""
    return importlib.import_module("anyfeature-vslam_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o  # CPU checker (test infrastructure)
    o.lib()
    return o


@pytest.fixture(scope="session")
def gpu_ctx(afv):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return afv.Context(max_width=1280, max_height=720, max_batch=8)
