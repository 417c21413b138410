"""Models resident in HBM (SLIMGPU_LearnResident / ModelFetch, include/slim_gpu.h): the grid loop of
src/programs/slim_mselect.c:94-113 without the model crossing PCIe twice per pair.  The bar: the fetched
host model is SLIM_Learn's (SaveModel, estimate.c:570-593), both views, bit for bit -- on every solver
path, cold and warm, with a fetch running beside the next solve, and when a column overflows its arena."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from slim_amd import _lib
from slim_amd.constants import SLIM_ERROR_INPUT, SLIM_OK
from slim_amd.engine import KERNEL_GRAM, KERNEL_TILE, KERNEL_WAVE_LDS, DeviceMatrix

pytestmark = pytest.mark.gpu


def views(lib, h):
    """(colptr, colind, colval, rowptr, rowind, rowval) of a host model handle, copied."""
    v = C.cast(h, C.POINTER(_lib.CsrView)).contents
    n = int(v.ncols)
    cp = np.ctypeslib.as_array(v.colptr, shape=(n + 1,)).copy()
    rp = np.ctypeslib.as_array(v.rowptr, shape=(n + 1,)).copy()
    nnz = int(cp[-1])
    assert int(rp[-1]) == nnz

    def arr(p):
        return np.ctypeslib.as_array(p, shape=(nnz,)).copy() if nnz else np.zeros(0)
    return cp, arr(v.colind), arr(v.colval), rp, arr(v.rowind), arr(v.rowval)


def same_model(lib, ha, hb):
    for a, b in zip(views(lib, ha), views(lib, hb)):
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)


def free(lib, h):
    lib.SLIM_FreeModel(C.byref(C.c_void_p(h)))


def ratings(nrows, ncols, density, seed, binary):
    rng = np.random.default_rng(seed)
    R = sp.random(nrows, ncols, density=density, format="csr", random_state=rng, dtype=np.float32)
    R.data[:] = 1.0 if binary else rng.integers(1, 6, R.nnz).astype(np.float32)
    return R


@pytest.mark.parametrize("kernel,shape,binary", [
    (KERNEL_WAVE_LDS, (900, 300, 0.05), False),
    (KERNEL_TILE, (30000, 700, 0.01), True),
    (KERNEL_GRAM, (30000, 700, 0.01), True),     # byte planes
    (KERNEL_GRAM, (20000, 500, 0.02), False),    # integer ratings
])
def test_resident_model_is_the_host_model_cold_and_warm(kernel, shape, binary, monkeypatch):
    # (a warm step from a resident model may start from the g the previous solve left instead of folding
    # the model again -- rounding-level differences, tested below; here the fold, which is SLIM_Learn's)
    monkeypatch.setenv("SLIM_GPU_NO_CARRY", "1")
    R = ratings(*shape, seed=5, binary=binary)
    mat = DeviceMatrix.from_scipy(R, binary=binary)
    lib = mat._lib
    kw = dict(optTol=1e-7, niters=200, seed=3, kernel=kernel)
    h1, _ = mat.learn(return_handle=True, l1r=2.0, l2r=1.0, **kw)
    d1, st = mat.learn_resident(l1r=2.0, l2r=1.0, **kw)
    assert d1.nnz == st["nnzW"] > 0
    f1 = d1.fetch(return_handle=True)
    same_model(lib, h1, f1)
    # warm: host handle vs resident model, then the next pair from each
    h2, s2h = mat.learn(imodel=h1, return_handle=True, l1r=2.0, l2r=5.0, **kw)
    d2, s2d = mat.learn_resident(warm=d1, l1r=2.0, l2r=5.0, **kw)
    assert s2h["sweeps"] == s2d["sweeps"] and s2h["D"] == s2d["D"]
    d2.fetch_begin()                       # copies while the next solve runs
    d3, _ = mat.learn_resident(warm=d2, l1r=1.0, l2r=5.0, **kw)
    f2 = d2.fetch(return_handle=True)
    same_model(lib, h2, f2)
    h3, _ = mat.learn(imodel=h2, return_handle=True, l1r=1.0, l2r=5.0, **kw)
    f3 = d3.fetch(return_handle=True)
    same_model(lib, h3, f3)
    f3b = d3.fetch(return_handle=True)     # a second fetch copies again
    same_model(lib, f3, f3b)
    for h in (h1, h2, h3, f1, f2, f3, f3b):
        free(lib, h)
    for d in (d1, d2, d3):
        d.free()
    assert d1.nnz == -1


def test_resident_model_when_a_column_overflows_its_arena(monkeypatch):
    """Columns that do not fit the output arena are solved again in a further launch, whose tiles are
    regrouped (which columns overflow depends on the order the launch finishes them in: such a model is
    reproducible to the solver's tolerance, not to the bit).  The resident model then gathers the pieces
    of several launches: both of its views must be the same matrix, well-formed, and the host solve's
    to the parity tolerance."""
    R = ratings(20000, 400, 0.02, seed=9, binary=True)
    mat = DeviceMatrix.from_scipy(R, binary=True)
    lib = mat._lib
    kw = dict(l1r=0.5, l2r=1.0, optTol=1e-11, niters=2000, seed=1, kernel=KERNEL_TILE)   # (tight: two tile groupings)
    W, st = mat.learn(**kw)
    monkeypatch.setenv("SLIM_GPU_ARENA", str(max(1024, int(st["nnzW"]) // 3)))
    d, st2 = mat.learn_resident(**kw)
    f = d.fetch(return_handle=True)
    cp, ci, cv, rp, ri, rv = views(lib, f)
    n = W.shape[0]
    assert cp[-1] == rp[-1] == st2["nnzW"] == d.nnz and abs(int(cp[-1]) - W.nnz) <= 5
    Wc = sp.csc_matrix((cv, ci, cp), shape=(n, n))
    Wr = sp.csr_matrix((rv, ri, rp), shape=(n, n))
    assert Wc.has_sorted_indices and Wr.has_sorted_indices
    for k in range(n):   # ids strictly ascending in every column and row
        assert np.all(np.diff(ci[cp[k]:cp[k + 1]]) > 0) and np.all(np.diff(ri[rp[k]:rp[k + 1]]) > 0)
    assert abs(Wc - Wr.tocsc()).nnz == 0
    dW = abs(Wc - W)
    assert (dW.max() if dW.nnz else 0.0) <= 2e-5
    free(lib, f)


def test_resident_model_errors():
    R = ratings(2000, 200, 0.05, seed=2, binary=True)
    mat = DeviceMatrix.from_scipy(R, binary=True)
    lib = mat._lib
    st = C.c_int32(0)
    assert not lib.SLIMGPU_ModelFetch(None, C.byref(st)) and st.value == SLIM_ERROR_INPUT
    assert lib.SLIMGPU_ModelFetchBegin(None) == SLIM_ERROR_INPUT
    lib.SLIMGPU_ModelFree(C.byref(C.c_void_p(None)))   # no-op
    d, _ = mat.learn_resident(l1r=1.0, l2r=1.0)
    W = d.fetch()
    W2, _ = mat.learn(l1r=1.0, l2r=1.0)
    assert abs(W - W2).nnz == 0 or abs(W - W2).max() == 0


def test_gram_builder_in_user_passes_forms_the_same_G(monkeypatch):
    """G = R^T R of a binary matrix with more users than 32 cluster members hold as one word each in
    LDS (C5: 10M users) is formed in user passes (engine.hip: gram_passes) -- the same G, hence the
    same item-space models bit for bit, as the single-launch form and as the line-gathering form."""
    R = ratings(40000, 600, 0.01, seed=11, binary=True)
    kw = dict(l1r=1.0, l2r=1.0, optTol=1e-7, niters=100, seed=1, kernel=KERNEL_GRAM)
    W0, s0 = DeviceMatrix.from_scipy(R, binary=True).learn(**kw)          # one launch, words in LDS
    monkeypatch.setenv("SLIM_GPU_TEST_HOOKS", "1")
    monkeypatch.setenv("SLIM_GPU_TEST_GBITS_ROWS", "300")                 # 1250 users per member -> 5 passes
    W1, s1 = DeviceMatrix.from_scipy(R, binary=True).learn(**kw)
    monkeypatch.delenv("SLIM_GPU_TEST_GBITS_ROWS")
    monkeypatch.setenv("SLIM_GPU_NO_GBITS", "1")                          # the line-gathering builder
    W2, s2 = DeviceMatrix.from_scipy(R, binary=True).learn(**kw)
    assert s0["nnzW"] == s1["nnzW"] == s2["nnzW"] > 0
    assert abs(W0 - W1).nnz == 0 and abs(W0 - W2).nnz == 0
    assert s0["sweeps"] == s1["sweeps"] == s2["sweeps"]


def test_predict_through_a_resident_model_equals_predict_on_the_fetched_one():
    R = ratings(3000, 400, 0.03, seed=4, binary=False)
    mat = DeviceMatrix.from_scipy(R)
    lib = mat._lib
    d, _ = mat.learn_resident(l1r=1.0, l2r=2.0)
    h = d.fetch(return_handle=True)
    Rc = sp.csr_matrix(R)
    ptr = np.ascontiguousarray(Rc.indptr, dtype=np.intp)
    ind = np.ascontiguousarray(Rc.indices, dtype=np.int32)
    val = np.ascontiguousarray(Rc.data, dtype=np.float32)
    trn = C.c_void_p()
    assert lib.Py_csr_wrapper(Rc.shape[0], ptr, ind, val.ctypes.data_as(C.c_void_p), C.byref(trn)) == SLIM_OK
    n = 10
    out = [np.full(Rc.shape[0] * n, -1, np.int32) for _ in range(2)]
    sc = [np.zeros(Rc.shape[0] * n, np.float32) for _ in range(2)]
    assert lib.SLIMGPU_ModelPredict(n, d.handle, trn, out[0].ctypes.data_as(C.c_void_p),
                                    sc[0].ctypes.data_as(C.c_void_p)) == SLIM_OK
    assert lib.SLIMGPU_Predict(n, C.c_void_p(h), trn, out[1], sc[1]) == SLIM_OK
    assert np.array_equal(out[0], out[1]) and np.array_equal(sc[0], sc[1]) and (out[0] >= 0).any()
    free(lib, h)


def test_g_carried_from_pair_to_pair_gives_the_folded_models(monkeypatch):
    """A grid step that only moves l2 starts from the g the previous solve left on chip (cd_gramr.hpp:
    g_save / g_load) instead of folding the previous model into g row by row (cd.c:108-110 in item
    space) -- the same quantity up to fp32 rounding: the models of a chain of pairs agree with the
    folded chain's to the parity tolerance, with the same sweep counts; a step that moves l1 folds."""
    R = ratings(30000, 700, 0.01, seed=21, binary=True)
    chain = [(2.0, 1.0), (2.0, 3.0), (2.0, 0.5), (1.0, 0.5), (1.0, 2.0)]
    kw = dict(optTol=1e-7, niters=300, seed=3, kernel=KERNEL_GRAM)

    def run(carry):
        if carry:
            monkeypatch.delenv("SLIM_GPU_NO_CARRY", raising=False)
        else:
            monkeypatch.setenv("SLIM_GPU_NO_CARRY", "1")
        mat = DeviceMatrix.from_scipy(R, binary=True)
        prev, out = None, []
        for l1, l2 in chain:
            cur, st = mat.learn_resident(warm=prev, l1r=l1, l2r=l2, **kw)
            out.append((cur.fetch(), st["sweeps"], st["gram_rows"]))
            if prev is not None:
                prev.free()
            prev = cur
        return out
    folded, carried = run(False), run(True)
    for k, ((Wf, sf, rf), (Wc, sc, rc)) in enumerate(zip(folded, carried)):
        d = abs(Wf - Wc)
        assert (d.max() if d.nnz else 0.0) <= 2e-5, k
        assert abs(sf - sc) <= 0.01 * sf
        if k in (1, 2, 4):     # only l2 moved: the rows of the fold (one per kept coefficient) are not read
            assert rc <= rf - int(0.9 * folded[k - 1][0].nnz)
        elif k == 0:           # the cold pair streams what the folded chain's streams
            assert rc == rf
        else:                  # l1 moved: folds like the other chain (from a model a rounding apart)
            assert abs(rc - rf) <= 0.01 * rf


def test_mselect_on_the_item_space_path_with_resident_models_and_the_carried_g(monkeypatch, capfd):
    """Py_SLIM_Mselect end to end on a matrix the engine solves in item space (the grid announced): the
    models stay in HBM, the l2 steps start from the carried g, top-N lists come from the resident row
    view.  Against the same grid with host models (SLIM_GPU_RESIDENT=0: SLIM_Learn + Py_SLIM_Predict per
    pair): the same best cells, HR / ARHR to 1e-3 (a list can change where two scores are a rounding
    apart), nnz of every pair within 0.1 %."""
    import re
    from slim_amd import SLIM, SLIMatrix
    rng = np.random.default_rng(31)
    R = sp.random(40000, 2500, density=0.01, format="csr", random_state=rng, dtype=np.float32)
    R.data[:] = 1.0
    R.sort_indices()
    # leave one rating per user out (users with at least two)
    trn, tst = R.tolil(copy=True), sp.lil_matrix(R.shape, dtype=np.float32)
    for u in range(R.shape[0]):
        cols = R.indices[R.indptr[u]:R.indptr[u + 1]]
        if cols.size >= 2:
            j = int(cols[rng.integers(cols.size)])
            trn[u, j] = 0
            tst[u, j] = 1.0
    trn, tst = sp.csr_matrix(trn), sp.csr_matrix(tst)
    trn.eliminate_zeros()
    params = {"dbglvl": 0, "algo": "cd", "nthreads": 1, "optTol": 1e-7, "niters": 200}
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SLIM_GPU_RESIDENT", mode)
        trainmat = SLIMatrix(trn)
        valmat = SLIMatrix(tst, trainmat)
        model = SLIM()
        C.CDLL(None).fflush(None)            # (whatever earlier tests left in libc's buffer is not this grid's)
        capfd.readouterr()
        model.mselect(params, trainmat, valmat, [1.0, 2.0], [1.0, 5.0, 10.0], nrcmds=10)
        C.CDLL(None).fflush(None)            # (the library prints through libc's buffered stdout)
        out = capfd.readouterr().out
        lines = re.findall(r"l1r: (\S+) l2r: (\S+) nnz:\s+(\d+) hr: (\S+) hr_head: \S+ hr_tail: \S+ arhr: (\S+)", out)
        assert len(lines) == 6
        st = _lib.Stats()
        model._lib.SLIMGPU_LastStats(C.byref(st))
        res[mode] = (model.mselect_result, lines, st.kernel, st.gram_rows, st.nnzW)
    (ra, la, ka, rows_a, nnz_a), (rb, lb, kb, rows_b, nnz_b) = res["1"], res["0"]
    assert ka == kb == KERNEL_GRAM
    assert ra["bestHR"][:2] == rb["bestHR"][:2] and ra["bestAR"][:2] == rb["bestAR"][:2]
    for x, y in zip(la, lb):
        assert x[:2] == y[:2] and abs(int(x[2]) - int(y[2])) <= 1e-3 * int(y[2])
        assert abs(float(x[3]) - float(y[3])) <= 1e-3 and abs(float(x[4]) - float(y[4])) <= 1e-3
    # the last pair is an l2 step: folded, it reads a row per coefficient of the previous model before its
    # sweeps; from the carried g only the sweeps' rows
    assert rows_b - rows_a >= 0.95 * int(lb[4][2]) and rows_a < rows_b
