import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def gpu_decoder():
    import rtlsdr_ft8d_amd as ft8
    dec = ft8.Decoder(device=0, max_frames=16)
    yield dec
    dec.close()
