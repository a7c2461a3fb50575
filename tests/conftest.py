import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# libwsa reads its tuning / test switches (WSA_DBG, WSA_NO_PAIR, ...: tools/README.md) only when this is set: the equivalence tests
# compare kernel variants through them; a host process that embeds the library never sees them
os.environ.setdefault("WSA_TUNING_ENV", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference + node (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
