import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the library is a build artefact (git-ignored): if this checkout has none yet, build it the way build() does
    lib = os.path.join(ROOT, "hybrid-drt_amd", "libhipdrt.so")
    if not os.path.exists(lib) and "HIPDRT_LIB" not in os.environ:
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
