import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# (the suite runs with the library's DEFAULT environment; knobs are exercised in processes of their own: tests/test_gpu_env_knobs.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = ast.literal_eval(str(z["__meta__"]))
    return z, meta


@pytest.fixture(scope="session")
def golden():
    return load_golden
