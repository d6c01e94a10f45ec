import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def drand48(n, seed=42):
    """The charge vector of SURVEY.md section 8(d) config 1(ii): srand48(42); drand48() in panel order."""
    libc = ctypes.CDLL("libc.so.6")
    libc.drand48.restype = ctypes.c_double
    libc.srand48(seed)
    return np.array([libc.drand48() for _ in range(n)])


def rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b))


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def fb():
    import fmm_bem_relaxed_amd
    return fmm_bem_relaxed_amd


@pytest.fixture(scope="session")
def gpu_available():
    import torch
    return torch.cuda.is_available()
