import os
import sys

import numpy as np
import pytest

TESTS = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(TESTS, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    with open(os.path.join(GOLDEN, "INDEX.txt")) as fh:
        return [l.strip() for l in fh if l.strip()]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    from _oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from _oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libvelox_ref.so not built (needs /root/reference)")
    return Reference(threads=8)


@pytest.fixture(scope="session")
def ctx():
    from velox_amd.codec import Context
    return Context(0)
