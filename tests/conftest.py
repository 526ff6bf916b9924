import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name is not an identifier)."""
    return importlib.import_module("go-raytracer_amd")


@pytest.fixture(scope="session")
def ob():
    """oracle/binding.py — the float64 CPU restatement (test infrastructure)."""
    from oracle import binding

    binding.load()
    return binding


GOLDEN = os.path.join(ROOT, "tests", "golden")
