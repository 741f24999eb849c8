import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU is a configuration error, not a pass: fail loudly there.
    # Without -m, GPU tests are skipped when no GPU is visible.
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    if not _gpu_available():
        skip = pytest.mark.skip(reason="no GPU visible")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def ref():
    import oracle_lib
    lib = oracle_lib.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return lib


@pytest.fixture(scope="session")
def avr():
    import avrecode_ms_amd
    avrecode_ms_amd.lib()
    return avrecode_ms_amd


@pytest.fixture
def hooks(avr):
    """hooks(name=value, ...): for the rest of the test, avr.lib() is the -DAVR_TEST_HOOKS build of the library
    (libavrecode_hip_hooks.so) with these hooks set; the product library has no such switches.  A second call replaces
    the first one's settings."""
    import contextlib
    stack = contextlib.ExitStack()

    def set_hooks(**kw):
        stack.close()
        stack.enter_context(avr.test_hooks(**kw))
    yield set_hooks
    stack.close()
