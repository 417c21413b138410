"""C-ABI of libslim.so: every symbol declared in include/*.h is exported with the
reference's calling convention; the host-side entry points (handles, top-N, head/tail,
model files) agree with the oracle.  No GPU needed: nothing here computes a CD update."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import slim_oracle as O
from conftest import ROOT, has_gpu
from slim_amd import _lib
from slim_amd.constants import SLIM_ERROR_INPUT, SLIM_NOPTIONS, SLIM_OK


def _declared():
    names = set()
    for h in ("slim.h", "slim_gpu.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b((?:SLIM|Py|SLIMGPU)_\w+)\s*\(", text))
    return names


def test_every_declared_symbol_is_exported():
    lib = C.CDLL(_lib.LIB_PATH)
    declared = _declared()
    assert len(declared) == 46
    for name in declared:
        assert hasattr(lib, name), "libslim.so does not export %s" % name
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    # the reference's Python wrapper resolves exactly these by name (core.py:366-385, 692-804)
    for name in ("Py_csr_wrapper", "Py_csr_free", "Py_SLIM_Learn", "Py_SLIM_Mselect",
                 "Py_SLIM_Predict", "Py_SLIM_Predict_1vsk", "Py_csr_save", "Py_csr_load",
                 "Py_csr_stat", "Py_csr_export"):
        assert name in declared


def test_handle_layout_matches_gk_csr_t():
    # 2 x int32, then 22 pointers (gk_csr_t field order, slim_gpu.h)
    assert C.sizeof(_lib.CsrView) == 8 + 22 * 8
    assert _lib.CsrView.rowptr.offset == 8 and _lib.CsrView.colptr.offset == 16
    assert _lib.CsrView.rowind.offset == 24 and _lib.CsrView.colind.offset == 32
    assert _lib.CsrView.rowval.offset == 88 and _lib.CsrView.colval.offset == 96
    assert _lib.CsrView.cnorms.offset == 112


def test_set_defaults():
    lib = _lib.load()
    io = np.zeros(SLIM_NOPTIONS, np.int32)
    do = np.zeros(SLIM_NOPTIONS, np.float64)
    assert lib.SLIM_iSetDefaults(io) == SLIM_OK and lib.SLIM_dSetDefaults(do) == SLIM_OK
    assert (io == -1).all() and (do == -1).all()


def _wrap(lib, M):
    M = sp.csr_matrix(M)
    h = C.c_void_p()
    val = np.ascontiguousarray(M.data, np.float32)
    rc = lib.Py_csr_wrapper(M.shape[0], np.ascontiguousarray(M.indptr, np.intp),
                            np.ascontiguousarray(M.indices, np.int32),
                            val.ctypes.data_as(C.c_void_p), C.byref(h))
    assert rc == SLIM_OK
    return h


def test_csr_wrapper_export_roundtrip(ml100k):
    lib = _lib.load()
    R, _ = ml100k
    h = _wrap(lib, R)
    view = C.cast(h, C.POINTER(_lib.CsrView)).contents
    assert view.nrows == 934 and view.ncols == 1683  # max id + 1 (setup.c:117)
    nnz = C.c_int32()
    assert lib.Py_csr_stat(h, C.byref(nnz)) == SLIM_OK and nnz.value == R.nnz
    ip = np.zeros(R.shape[0] + 1, np.int32)
    ix = np.zeros(R.nnz, np.int32)
    dv = np.zeros(R.nnz, np.float32)
    assert lib.Py_csr_export(h, ip, ix, dv) == SLIM_OK
    assert np.array_equal(ip, R.indptr) and np.array_equal(ix, R.indices)
    assert np.array_equal(dv, R.data)
    assert lib.Py_csr_free(h) == SLIM_OK


@pytest.fixture(scope="module")
def oracle_model(ml100k):
    R, T = ml100k
    W = O.learn_cd(R, order=O.ORDER_PERM, seed=1, aty=O.ATY_GRAM, nthreads=4)
    return W


def _model_handle(lib, W):
    from slim_amd.engine import _scipy_to_model_handle
    return _scipy_to_model_handle(lib, W)


def test_predict_matches_oracle(ml100k, oracle_model):
    lib = _lib.load()
    R, T = ml100k
    hm = _model_handle(lib, oracle_model)
    hr = _wrap(lib, R)
    out = np.full(R.shape[0] * 10, -1, np.int32)
    sc = np.zeros(R.shape[0] * 10, np.float32)
    assert lib.Py_SLIM_Predict(10, hm, hr, out, sc) == SLIM_OK
    ids, scores = O.predict(oracle_model, R, 10)
    assert np.array_equal(out.reshape(-1, 10), ids)
    assert np.array_equal(sc.reshape(-1, 10), scores)  # same float accumulation order
    # single-profile entry points
    lo, hi = R.indptr[5], R.indptr[6]
    items = np.ascontiguousarray(R.indices[lo:hi], np.int32)
    vals = np.ascontiguousarray(R.data[lo:hi], np.float32)
    rids = np.zeros(10, np.int32)
    rsc = np.zeros(10, np.float32)
    n = lib.SLIM_GetTopN(hm, items.size, items, vals.ctypes.data_as(C.c_void_p), None, 10, rids, rsc)
    assert n == 10 and np.array_equal(rids, ids[5]) and np.array_equal(rsc, scores[5])
    n = lib.Py_SLIM_GetTopN(hm, items.size, items, vals.ctypes.data_as(C.c_void_p), 10, rids, rsc, 0)
    assert n == 10 and np.array_equal(rids, ids[5])
    # NULL ratings = implicit feedback (predict.c:45)
    n = lib.SLIM_GetTopN(hm, items.size, items, None, None, 10, rids, rsc)
    o_ids, o_sc = O.get_topn(oracle_model, items, None, 10)
    assert n == 10 and np.array_equal(rids, o_ids) and np.array_equal(rsc, o_sc)
    # history is never recommended
    assert not set(rids.tolist()) & set(items.tolist())
    lib.Py_csr_free(hr)
    lib.SLIM_FreeModel(C.byref(hm))
    assert not hm.value


def test_predict_1vsk(ml100k, oracle_model):
    lib = _lib.load()
    R, _ = ml100k
    hm = _model_handle(lib, oracle_model)
    Wr = sp.csr_matrix(oracle_model)
    lo, hi = R.indptr[3], R.indptr[4]
    items = np.ascontiguousarray(R.indices[lo:hi], np.int32)
    vals = np.ascontiguousarray(R.data[lo:hi], np.float32)
    negs = np.array([10, 50, 100, 200, 300, 400, 1682, -1], np.int32)
    rids = np.zeros(5, np.int32)
    rsc = np.zeros(5, np.float32)
    n = lib.Py_SLIM_GetTopN_1vsk(hm, items.size, items, vals.ctypes.data_as(C.c_void_p), 5, rids,
                                 rsc, negs.size, negs, 0)
    assert n == 5
    full = np.zeros(oracle_model.shape[0], np.float32)
    for i, v in zip(items, vals):  # predict.c:106-119, float accumulation in history order
        row = Wr.getrow(i)
        full[row.indices] += np.float32(v) * row.data
    want = sorted(((full[k] if k >= 0 else 0.0, -j) for j, k in enumerate(negs)), reverse=True)[:5]
    assert np.allclose(rsc, [w[0] for w in want], rtol=0, atol=0)
    assert [int(negs[-w[1]]) for w in want] == rids.tolist()
    lib.SLIM_FreeModel(C.byref(hm))


def test_head_tail_matches_oracle(ml100k):
    lib = _lib.load()
    R, _ = ml100k
    ptr = np.ascontiguousarray(R.indptr, np.intp)
    ind = np.ascontiguousarray(R.indices, np.int32)
    p = lib.SLIM_DetermineHeadAndTail(R.shape[0], 1683, ptr, ind)
    got = np.ctypeslib.as_array(p, shape=(1683,)).copy()
    C.CDLL(None).free(p)
    want = O.head_tail(R, 1683)
    assert np.array_equal(got, want)
    # definition (api.c:236-241): the head covers at least half of the ratings
    pop = np.bincount(R.indices, minlength=1683)
    assert pop[got == 0].sum() >= R.nnz // 2
    assert pop[got == 0].min() >= pop[got == 1].max()


def test_model_files_roundtrip(tmp_path, oracle_model):
    lib = _lib.load()
    hm = _model_handle(lib, oracle_model)
    Wr = sp.csr_matrix(oracle_model)
    Wr.sort_indices()
    # binary row format (api.c:174-194): int32 nrows, int32 ncols, ssize_t rowptr[], ...
    path = str(tmp_path / "w.bin").encode()
    assert lib.SLIM_WriteModel(hm, path) == SLIM_OK
    raw = open(path, "rb").read()
    assert np.frombuffer(raw[:8], np.int32).tolist() == [1683, 1683]
    assert len(raw) == 8 + 8 * 1684 + 8 * Wr.nnz
    h2 = C.c_void_p(lib.SLIM_ReadModel(path))
    v2 = C.cast(h2, C.POINTER(_lib.CsrView)).contents
    assert v2.colptr and v2.rowptr  # api.c:191 adds the column view
    assert np.array_equal(np.ctypeslib.as_array(v2.rowptr, shape=(1684,)), Wr.indptr)
    assert np.array_equal(np.ctypeslib.as_array(v2.rowval, shape=(Wr.nnz,)), Wr.data)
    Wc = sp.csc_matrix(oracle_model)
    Wc.sort_indices()
    assert np.array_equal(np.ctypeslib.as_array(v2.colind, shape=(Wr.nnz,)), Wc.indices)
    # text CSR (pyapi.c:47-64)
    tpath = str(tmp_path / "w.csr").encode()
    assert lib.Py_csr_save(hm, tpath) == SLIM_OK
    h3 = C.c_void_p()
    assert lib.Py_csr_load(C.byref(h3), tpath) == SLIM_OK
    v3 = C.cast(h3, C.POINTER(_lib.CsrView)).contents
    assert v3.nrows == 1683
    assert np.array_equal(np.ctypeslib.as_array(v3.rowind, shape=(Wr.nnz,)), Wr.indices)
    assert np.array_equal(np.ctypeslib.as_array(v3.rowval, shape=(Wr.nnz,)), Wr.data)  # lossless
    for h in (hm, h2, h3):
        lib.SLIM_FreeModel(C.byref(h))


def test_bad_arguments_are_errors_not_crashes():
    lib = _lib.load()
    rids = np.zeros(3, np.int32)
    rsc = np.zeros(3, np.float32)
    items = np.zeros(1, np.int32)
    assert lib.SLIM_GetTopN(None, 1, items, None, None, 3, rids, rsc) < 0
    assert lib.Py_csr_stat(None, C.byref(C.c_int32())) < 0
    h = C.c_void_p()
    assert lib.Py_csr_load(C.byref(h), b"/nonexistent/file") < 0
    assert not lib.SLIM_ReadModel(b"/nonexistent/file")
    lib.SLIM_FreeModel(C.byref(C.c_void_p()))  # NULL is fine
    # models resident in HBM (slim_gpu.h): a null matrix / model is an input error, a null free a no-op
    st = C.c_int32(12345)
    assert not lib.SLIMGPU_LearnResident(None, None, None, None, C.byref(st)) and st.value == SLIM_ERROR_INPUT
    assert "null matrix" in _lib.last_error()
    st = C.c_int32(12345)
    assert not lib.SLIMGPU_ModelFetch(None, C.byref(st)) and st.value == SLIM_ERROR_INPUT
    assert lib.SLIMGPU_ModelFetchBegin(None) == SLIM_ERROR_INPUT and lib.SLIMGPU_ModelNnz(None) == -1
    assert lib.SLIMGPU_ModelPredict(10, None, None, None, None) == SLIM_ERROR_INPUT
    lib.SLIMGPU_ModelFree(C.byref(C.c_void_p()))


@pytest.mark.skipif(has_gpu(), reason="this box has a GPU")
def test_training_fails_loudly_without_a_gpu(ml100k):
    """No CPU fallback: SLIM_Learn reports an error instead of computing on the host."""
    lib = _lib.load()
    R, _ = ml100k
    st = C.c_int32(12345)
    val = np.ascontiguousarray(R.data, np.float32)
    h = lib.SLIM_Learn(R.shape[0], np.ascontiguousarray(R.indptr, np.intp),
                       np.ascontiguousarray(R.indices, np.int32), val.ctypes.data_as(C.c_void_p),
                       None, None, None, C.byref(st))
    assert not h and st.value < 0
    assert "no CPU fallback" in _lib.last_error()
    from slim_amd import SLIM, SLIMatrix
    with pytest.raises(RuntimeError):
        SLIM().train({}, SLIMatrix(R))


def test_unsupported_algorithms_are_input_errors(ml100k):
    """api.c:81-84 prints "Algorithm not supported" and exits; here: SLIM_ERROR_INPUT."""
    lib = _lib.load()
    R, _ = ml100k
    io = np.full(SLIM_NOPTIONS, -1, np.int32)
    io[5] = 7                                       # SLIM_OPTION_ALGO: neither admm (0) nor cd (1)
    st = C.c_int32(0)
    val = np.ascontiguousarray(R.data, np.float32)
    h = lib.SLIM_Learn(R.shape[0], np.ascontiguousarray(R.indptr, np.intp),
                       np.ascontiguousarray(R.indices, np.int32), val.ctypes.data_as(C.c_void_p),
                       io.ctypes.data_as(C.c_void_p), None, None, C.byref(st))
    assert not h and st.value == -2
    assert "unknown algorithm" in _lib.last_error()


def test_large_model_row_view_is_built_on_several_threads():
    """A model of more than 4M entries takes the threaded counting-sort transpose of
    host_csr.cpp::csr_build_index (a C5 grid step returns 77M entries, 45 times over): the row
    view must be the one scipy builds -- same offsets, ids ascending inside every row, same values."""
    import ctypes as C
    import scipy.sparse as sp
    from slim_amd.engine import _scipy_to_model_handle, model_to_scipy
    lib = _lib.load()
    rng = np.random.default_rng(0)
    W = sp.random(3000, 3000, density=0.6, format="csr", random_state=rng, dtype=np.float32)
    W.sort_indices()
    h = _scipy_to_model_handle(lib, W)
    view = C.cast(h, C.POINTER(_lib.CsrView)).contents
    n = int(view.ncols)
    rp = np.ctypeslib.as_array(view.rowptr, shape=(n + 1,)).astype(np.int64)
    nnz = int(rp[-1])
    assert nnz == W.nnz > (1 << 22)
    ri = np.ctypeslib.as_array(view.rowind, shape=(nnz,)).copy()
    rv = np.ctypeslib.as_array(view.rowval, shape=(nnz,)).copy()
    assert np.array_equal(rp, W.indptr) and np.array_equal(ri, W.indices) and np.array_equal(rv, W.data)
    Wc = model_to_scipy(lib, h, free=True)
    assert abs(Wc.tocsr() - W).max() == 0.0
