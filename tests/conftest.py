import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the builder's experiment builds (tools/bin/*.so): BNMF_TEST_LIB points the whole suite at another build of the library
if os.environ.get("BNMF_TEST_LIB"):
    import bayesnmf_amd.engine as _E
    _E.LIB_PATH = os.path.abspath(os.environ["BNMF_TEST_LIB"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle
