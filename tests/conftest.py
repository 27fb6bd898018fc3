import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build what is missing (oracle restatement, input generator); the HIP library is built by
    __graft_entry__.build() / make and is only required by the tests that use it."""
    import subprocess

    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(os.path.join(ROOT, "md_neighbor_list_amd", "lib", "libnl_inputs.so")):
        subprocess.check_call(["make", "-C", ROOT, "inputs"])
    yield
