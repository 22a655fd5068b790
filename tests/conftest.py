import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def ascii_set():
    import fixtures
    return fixtures.load_ascii()


@pytest.fixture(scope="session")
def ctx():
    """one fr_ctx for the GPU session (a single process on the card)"""
    import font_renderer_amd as fr
    c = fr.Context(0)
    yield c
    c.close()
