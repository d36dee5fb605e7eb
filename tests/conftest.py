import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs oracle/_ref binaries built from /root/reference")


@pytest.fixture(scope="session")
def engine():
    """One PoaEngine on cuda:0 for the whole session; fails loudly without the HIP library."""
    from elector_amd.poa import PoaEngine
    eng = PoaEngine(0)
    yield eng
    eng.close()
