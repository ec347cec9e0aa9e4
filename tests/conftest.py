import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TOOLS = os.path.join(ROOT, "tests", "tools")
if TOOLS not in sys.path:
    sys.path.insert(0, TOOLS)

# the library's fault-injection option (prover_fail_round) exists only in a process started with test hooks on (bbg_capi.hip): this is one
os.environ.setdefault("BBG_TEST_HOOKS", "1")

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference build; only where oracle/_ref/libbbref.so exists and the CPU can run it."""
    from oracle.oracle import Ref, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref/libbbref.so not available here")
    return Ref()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def bbg(pkg):
    """GPU context through the C ABI.  Fails loudly (no fallback) when the extension or the GPU is missing."""
    ctx = pkg.Bbg(0)
    # several tests stage device buffers with torch (zeros, clones) and then hand them to the library: run both on torch's
    # current stream so that a torch fill can never race a library kernel on the same buffer
    import torch
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    yield ctx
    ctx.close()


def unhex(s, last=4):
    return np.frombuffer(bytes.fromhex(s), dtype=np.uint64).reshape(-1, last).copy()


def limbs(lst):
    return np.array([int(x, 16) for x in lst], dtype=np.uint64)


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()
