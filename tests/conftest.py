import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")
MODELS = os.path.join(ROOT, "assets", "models")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by the reference's own code (oracle/gen_golden.py)."""
    return dict(np.load(GOLDEN))


@pytest.fixture(scope="session")
def benchmark_golden():
    """The benchmark's own scenes rendered by the reference's code (oracle/gen_benchmark_golden.py)."""
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "benchmark_scenes.npz")))


BENCHMARK_ROWS_1080P = (137, 540, 803)


def sha256_of(array):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(array).tobytes()).digest(), np.uint8)


@pytest.fixture(scope="session")
def orc():
    """The CPU restatement (oracle/libpt_oracle.so): the checker, never the product."""
    from oracle.bindings import OracleLib, have_oracle
    if not have_oracle():
        pytest.fail("oracle/libpt_oracle.so is missing: run `make` (or __graft_entry__.build())")
    return OracleLib()


@pytest.fixture(scope="session")
def ref():
    """The reference's own sources built in place (only where that build exists)."""
    from oracle.bindings import RefLib, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/libptref.so not built (needs the reference tree)")
    return RefLib()


@pytest.fixture(scope="session")
def api():
    from pathtracer_amd import api as module
    module.lib()
    return module


def same_bits(a, b):
    """Bit-for-bit equality of two float arrays, except that any NaN equals any NaN (x86 and
    gfx950 produce default NaNs of opposite sign)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape:
        return False
    nan = np.isnan(a)
    if not np.array_equal(nan, np.isnan(b)):
        return False
    return np.array_equal(a[~nan].view(np.uint64), b[~nan].view(np.uint64))


def id_hash(indices):
    idx = indices[indices >= 0].astype(np.uint64)
    return int(((idx * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)).sum())
