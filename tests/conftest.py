import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# BLAS pools sized to the CPU share of this process (a GPU box may expose 128 cores but grant 16)
import psm_amd  # noqa: E402
_blas_limit = psm_amd.hostinfo.limit_blas_threads()
