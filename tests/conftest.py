import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    # SSP_TEST_OPT="name=value,...": run the suite under experiment knobs of the library (A/B correctness runs of a kernel
    # variant before it becomes the default); unset in the driver's runs
    opts = os.environ.get('SSP_TEST_OPT', '')
    if opts:
        from singleshotpose_amd import _lib
        for kv in filter(None, opts.split(',')):
            name, val = kv.split('=')
            _lib.call('ssp_set_option', name.encode(), int(val))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def root_dir():
    return ROOT
