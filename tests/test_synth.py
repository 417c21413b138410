"""The synthetic-matrix generator of the large benchmark configurations (slim_amd/synth.py,
SURVEY.md 8(d)): valid CSR, items drawn without replacement per user, the requested nnz hit,
reproducible from the seed, a popularity law that does not depend on the item id."""
import numpy as np
import torch

from slim_amd import synth


def _gen(nr, nc, nz, **kw):
    ptr, ind, val = synth.generate_csr(nr, nc, nz, device="cpu", **kw)
    return ptr.numpy(), ind.numpy(), val.numpy()


def test_valid_csr_without_duplicates_and_target_nnz():
    nr, nc, nz = 20000, 5000, 1_000_000
    ptr, ind, val = _gen(nr, nc, nz, seed=3)
    assert ptr[0] == 0 and ptr[-1] == ind.size == val.size
    assert abs(ind.size - nz) <= 0.01 * nz                    # nnz after merging
    assert ind.min() >= 0 and ind.max() < nc
    deg = np.diff(ptr)
    assert deg.min() >= 5 and deg.max() <= min(5000, nc // 2)
    # ascending, hence distinct, item ids inside every row
    inc = np.diff(ind.astype(np.int64)) > 0
    row_start = np.zeros(ind.size, bool)
    row_start[ptr[1:-1]] = True
    assert (inc | row_start[1:]).all()
    assert (val == 1.0).all()


def test_reproducible_and_seed_dependent():
    a = _gen(3000, 800, 60000, seed=1)
    b = _gen(3000, 800, 60000, seed=1)
    c = _gen(3000, 800, 60000, seed=2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert not np.array_equal(a[1][:1000], c[1][:1000])


def test_popularity_is_skewed_but_not_tied_to_the_id():
    nr, nc, nz = 30000, 2000, 900000
    ptr, ind, _ = _gen(nr, nc, nz, seed=7)
    pop = np.bincount(ind, minlength=nc).astype(np.float64)
    top = np.sort(pop)[::-1]
    assert top[:nc // 100].sum() > 2.5 * top[-nc // 100:].sum() * 1.0   # head >> tail
    assert abs(np.corrcoef(np.arange(nc), pop)[0, 1]) < 0.1            # ids are a random relabelling


def test_ratings_variant_and_scaled_configs():
    ptr, ind, val = _gen(2000, 500, 30000, seed=1, ratings=True)
    assert set(np.unique(val)) <= {1.0, 2.0, 3.0, 4.0, 5.0} and (val == 5.0).mean() > 0.4
    nr, nc, nz = synth.scaled("c4", 0.01)
    assert (nr, nc) == (10000, 1000) and nz == 100000
    assert synth.CONFIGS["c4"] == (1_000_000, 100_000, 1_000_000_000)
