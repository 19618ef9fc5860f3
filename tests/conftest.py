import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def case_cache():
    """Cases are expensive to regenerate (N up to 1e7); build each once per session."""
    from tests import cases
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = cases.build_case(name)
        return cache[name]
    return get
