"""Shared pytest plumbing: the `gpu` marker, import paths and library handles.

CPU tests (`-m "not gpu"`) cover the oracle against the golden vectors, host logic and the
C-ABI surface; `-m gpu` tests are the parity tests proper and call through the C ABI.
"""
import glob
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def pt():
    """The product's Python host layer (ctypes over libptmi.so)."""
    mod = importlib.import_module("cuda-pathtracer_amd")
    mod.load_library()
    return mod


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    import oracle_py
    oracle_py.load_oracle()
    return oracle_py


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not os.path.basename(p).startswith(("camera", "nee_", "t3_")))


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
