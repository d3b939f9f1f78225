import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    variant = os.environ.get("AFV_TEST_LIB") or None
    if variant:  # before any test module can load the in-tree library
        importlib.import_module("anyfeature-vslam_amd")._lib.use_library(variant)


@pytest.fixture(scope="session")
def afv():
    """the product package (directory name has a hyphen).  AFV_TEST_LIB=path binds a variant build of the library for this test run
    (the poison build of tools/poison_build.py: csrc/afv_poison.h) - a switch of the TESTS; the package's loader ignores the environment."""
    pkg = importlib.import_module("anyfeature-vslam_amd")
    variant = os.environ.get("AFV_TEST_LIB") or None
    if variant:
        assert os.path.samefile(pkg._lib.LIB_PATH, variant), "AFV_TEST_LIB came too late: the library is already loaded"
    return pkg


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
