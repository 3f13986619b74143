import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# the opt-in column-split launch of aptai_gemm_bf16 is exercised by the GPU suite (bit-identical to the single launch:
# tests/test_gpu_gemm.py::test_split_column_launch_equals_the_single_launch); the library reads the variable at its first GEMM
os.environ.setdefault("APTAI_GEMM_SPLITN", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = ast.literal_eval(str(z["__meta__"]))
    return z, meta


@pytest.fixture(scope="session")
def golden():
    return load_golden
