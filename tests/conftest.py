import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# `-x` stops at the first failure: the newest GPU tests run FIRST, so that one red old test can never hide a round's new tests
# from the driver again (round 5: 34 tests sat behind a red one).  Within a file the order is unchanged.
_FILE_ORDER = ["test_gpu_round6", "test_gpu_round5", "test_gpu_round4", "test_gpu_switches", "test_gpu_round3"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _FILE_ORDER.index(name) if name in _FILE_ORDER else len(_FILE_ORDER)
    items.sort(key=rank)  # (stable)
