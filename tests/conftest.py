import os
import sys

import numpy as np
import pytest

# torch ships its own HIP runtime; it must initialise BEFORE libslim.so (linked against the
# system ROCm) is loaded into this process, or torch finds "no ROCm-capable device" later
# (the full-size tests generate their matrices on the GPU with torch).  bench.py imports in the
# same order.
try:
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
except Exception:  # pragma: no cover - torch is plumbing here; CPU tests do not need it
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """A GPU kernel that never returns cannot be interrupted by a signal: give every GPU test a
    hard (thread-method) timeout so a hang costs minutes, not the whole box."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    for item in gpu_items:
        if not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(300, method="thread"))
    # without a device every GPU test is a skip, not a failure, so that a plain `pytest tests`
    # on a CPU-only box tells a regression from a missing GPU
    if gpu_items and not has_gpu():
        skip = pytest.mark.skip(reason="no gfx950 device: GPU tests need a real MI355X")
        for item in gpu_items:
            item.add_marker(skip)


def _ensure_built():
    import subprocess
    so = os.path.join(ROOT, "slim_amd", "libslim.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-j8", "-C", os.path.join(ROOT, "slim_amd", "csrc")])


def has_gpu():
    try:
        _ensure_built()
        from slim_amd import _lib
        return _lib.load().SLIMGPU_DeviceCount() > 0
    except Exception:
        return False


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Tests run against the in-tree libslim.so and the compiled oracle; build both if
    a fresh checkout lacks them (hipcc cross-compiles without a GPU)."""
    _ensure_built()
    import slim_oracle
    slim_oracle.build()


@pytest.fixture(scope="session")
def ml100k():
    from slim_amd.io import read_csr_text
    R = read_csr_text(os.path.join(GOLDEN, "ml100k-train.csr"))
    T = read_csr_text(os.path.join(GOLDEN, "ml100k-test.csr"), nrows=R.shape[0])
    return R, T


@pytest.fixture(scope="session")
def automotive_triplets():
    from slim_amd.io import read_ijv
    trn = read_ijv(os.path.join(GOLDEN, "AutomotiveTrain.ijv"))
    tst = read_ijv(os.path.join(GOLDEN, "AutomotiveTest.ijv"))
    return trn, tst


def map_triplets(trn, tst):
    """The reference wrapper's id mapping (python-package/SLIM/core.py:289-351) done
    with plain numpy: dense ids in first-appearance order; test events outside the
    training maps are dropped."""
    import scipy.sparse as sp

    def first_seen(keys):
        table, names = {}, []
        for k in keys:
            if k not in table:
                table[k] = len(names)
                names.append(k)
        return table, names

    u2i, users = first_seen(trn[:, 0])
    i2i, items = first_seen(trn[:, 1])
    R = sp.csr_matrix((trn[:, 2], ([u2i[u] for u in trn[:, 0]], [i2i[i] for i in trn[:, 1]])),
                      shape=(len(users), len(items)))
    keep = [(u2i[u], i2i[i], v) for u, i, v in tst if u in u2i and i in i2i]
    keep = np.array(keep)
    T = sp.csr_matrix((keep[:, 2], (keep[:, 0].astype(int), keep[:, 1].astype(int))),
                      shape=R.shape)
    return R, T, np.array(users), np.array(items)


@pytest.fixture(scope="session")
def automotive(automotive_triplets):
    return map_triplets(*automotive_triplets)
