"""Pins of the CPU oracle (oracle/slim_oracle.c) against what the reference itself
recorded.  The reference ships no tests and cannot be built here (GKlib absent), so
the pins are (1) the only outputs it records -- python-package/UserGuide.ipynb:275-277,
the best-HR / best-AR lines of the Automotive 9x9 model selection -- and (2) the
values SURVEY.md 8(c) measured on the reference's own sources in this image
(single thread, libc rand() never seeded): exact W nnz, sum, max, loss, fit, HR, ARHR.
"""
import numpy as np
import pytest

import slim_oracle as O


def test_ml100k_matches_reference_probe(ml100k):
    R, T = ml100k
    assert R.shape == (934, 1683) and R.nnz == 98222
    W, st, err, obj = O.learn_cd(R, l1r=1.0, l2r=1.0, optTol=1e-7, maxniters=10000,
                                 nthreads=1, order=O.ORDER_GLIBC, srand=1,
                                 aty=O.ATY_FULLSCAN, return_stats=True)
    # W nnz 65 928, sum 2207.331184, max 0.672417, loss 2.29460e4,
    # fit 2.06490e4 -- these depend on the exact rand() sequence and arithmetic
    assert W.nnz == 65928
    assert abs(W.data.astype(np.float64).sum() - 2207.331184) < 5e-6
    assert abs(float(W.data.max()) - 0.672417) < 1e-6
    assert "%.5e" % obj == "2.29460e+04"
    assert "%.5e" % err == "2.06490e+04"
    # per-column statistics of the same run
    assert np.median(st["nacols"]) == 937 and st["nacols"].max() == 1491
    assert st["sweeps"].min() == 1 and np.median(st["sweeps"]) == 12 and st["sweeps"].max() == 38
    nnzw = np.diff(W.indptr)
    assert np.median(nnzw) == 40 and nnzw.max() == 86
    assert abs(st["G"].sum() - 1.99e7) < 1e5
    ev = O.evaluate(W, R, T)
    assert ev["nvalid"] == 934
    assert "%.4f" % ev["hr"] == "0.3191"   # 298 / 934
    assert "%.4f" % ev["arhr"] == "0.1504"
    assert round(ev["hr"] * 934) == 298


def test_gram_aty_equals_fullscan(ml100k):
    """The engine's Gram-column aTy and the reference's full scan (estimate.c:412-421)
    give the same active sets and, with the same visiting order, the same W."""
    R, _ = ml100k
    cols = np.arange(0, 1683, 7, dtype=np.int32)
    a = O.learn_cd(R, order=O.ORDER_PERM, seed=3, aty=O.ATY_FULLSCAN, cols=cols)
    b = O.learn_cd(R, order=O.ORDER_PERM, seed=3, aty=O.ATY_GRAM, cols=cols)
    assert (a != b).nnz == 0


def test_order_noise_envelope(ml100k):
    """Reference self-noise: another visiting order moves W by ~1e-3 at
    optTol 1e-7 and by < 1e-5 at a tight tolerance."""
    R, _ = ml100k
    cols = np.arange(0, 1683, 5, dtype=np.int32)
    a = O.learn_cd(R, order=O.ORDER_GLIBC, srand=1, cols=cols, aty=O.ATY_GRAM)
    b = O.learn_cd(R, order=O.ORDER_PERM, seed=1, cols=cols, aty=O.ATY_GRAM)
    assert 1e-6 < abs(a - b).max() < 3e-3
    a = O.learn_cd(R, order=O.ORDER_GLIBC, srand=1, cols=cols, aty=O.ATY_GRAM, optTol=1e-12,
                   maxniters=100000)
    b = O.learn_cd(R, order=O.ORDER_PERM, seed=1, cols=cols, aty=O.ATY_GRAM, optTol=1e-12,
                   maxniters=100000)
    assert abs(a - b).max() < 2e-5


def test_fp32_arithmetic_is_within_tolerance(ml100k):
    R, _ = ml100k
    cols = np.arange(0, 1683, 5, dtype=np.int32)
    a = O.learn_cd(R, order=O.ORDER_PERM, seed=1, cols=cols, aty=O.ATY_GRAM)
    b = O.learn_cd(R, order=O.ORDER_PERM, seed=1, cols=cols, aty=O.ATY_GRAM, fp32=True)
    assert abs(a - b).max() < 2e-5


def test_perm_is_a_permutation():
    for n in (1, 2, 3, 7, 64, 65, 937, 1491, 4096, 100003):
        key = O.perm_key(1, n, 5)
        seen = sorted(O.perm_index(p, n, key) for p in range(min(n, 5000))) if n > 5000 else \
            sorted(O.perm_index(p, n, key) for p in range(n))
        if n <= 5000:
            assert seen == list(range(n))
        else:
            assert len(set(seen)) == len(seen) and seen[-1] < n


def test_automotive_train_matches_reference_probe(automotive):
    """Automotive through the wrapper's id mapping, l1=l2=1, niters=100:
    2928 x 1835, 17 545 nnz, W nnz 84 323, sum 5220.359019."""
    R, T, users, items = automotive
    assert R.shape == (2928, 1835) and R.nnz == 17545 and T.nnz == 2928
    W = O.learn_cd(R, maxniters=100, order=O.ORDER_GLIBC, srand=1)
    assert W.nnz == 84323
    assert abs(W.data.astype(np.float64).sum() - 5220.359019) < 5e-5


@pytest.mark.timeout(600)
def test_automotive_mselect_matches_notebook(automotive):
    """python-package/UserGuide.ipynb:276-277 (the reference's only recorded results):
      best HR: l1 20, l2 0.1  -> HR 0.1404, AR 0.0654
      best AR: l1 20, l2 50   -> HR 0.1390, AR 0.0669
    Grid, warm start and selection rule as Py_SLIM_Mselect (pyapi.c:286-403)."""
    R, T, _, _ = automotive
    l1s = sorted([0.01, 0.1, 0.5, 1, 2, 4, 5, 10, 20])
    l2s = sorted([0.1, 0.5, 1, 2, 5, 10, 20, 30, 50])
    best_hr = (0.0, None)
    best_ar = (0.0, None)
    model = None
    first = True
    cells = {}
    for l1 in l1s:
        for l2 in l2s:
            model = O.learn_cd(R, l1r=l1, l2r=l2, maxniters=100, order=O.ORDER_GLIBC,
                               srand=1 if first else None, imodel=model)
            first = False
            ev = O.evaluate(model, R, T)
            cells[(l1, l2)] = (model.nnz, ev["hr"], ev["arhr"])
            if ev["hr"] > best_hr[0]:
                best_hr = (ev["hr"], (l1, l2, ev["hr"], ev["arhr"]))
            if ev["arhr"] > best_ar[0]:
                best_ar = (ev["arhr"], (l1, l2, ev["hr"], ev["arhr"]))
    line_hr = "l1: %.4f, l2:%.4f, HR:%.4f, AR:%.4f." % best_hr[1]
    line_ar = "l1: %.4f, l2:%.4f, HR:%.4f, AR:%.4f." % best_ar[1]
    assert line_hr == "l1: 20.0000, l2:0.1000, HR:0.1404, AR:0.0654."
    assert line_ar == "l1: 20.0000, l2:50.0000, HR:0.1390, AR:0.0669."
    # SURVEY.md 8(c) per-cell values of the same grid.  The survey's probe ran the grid
    # at an unknown position of libc's rand() stream (other solves preceded it in the
    # same process), so nnz can differ by a few pattern entries; HR/ARHR do not.
    assert abs(cells[(20, 0.1)][0] - 36526) <= 20
    assert abs(cells[(20, 50)][0] - 39340) <= 20
    assert abs(cells[(10, 1)][0] - 68729) <= 20
    assert "%.4f %.4f" % cells[(10, 1)][1:] == "0.1233 0.0600"
