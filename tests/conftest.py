import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    config.addinivalue_line("markers", "slow: takes more than ~30 s on 8 CPU cores")


@pytest.fixture(scope="session")
def reference():
    import ref_import
    if not ref_import.have_reference():
        pytest.skip("/root/reference not present (GPU box): Tier-A checks run in the dev container only")
    return ref_import.import_reference()
