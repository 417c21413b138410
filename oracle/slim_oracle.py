"""ctypes binding of oracle/slim_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (slim_amd/) never does.  See slim_oracle.c for the
provenance of every function (reference file:line).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libslim_oracle.so")

ORDER_GLIBC, ORDER_PERM, ORDER_LOCAL, ORDER_NONE = 0, 1, 2, 3
ATY_FULLSCAN, ATY_GRAM = 0, 1


class Cfg(C.Structure):
    _fields_ = [("l1r", C.c_double), ("l2r", C.c_double), ("optTol", C.c_double),
                ("maxniters", C.c_int32), ("nthreads", C.c_int32),
                ("order", C.c_int32), ("seed", C.c_uint32),
                ("aty", C.c_int32), ("fp32", C.c_int32),
                ("nnbrs", C.c_int32), ("simtype", C.c_int32),
                ("chunk", C.c_int32), ("tile_first", C.c_int32), ("tile_count", C.c_int32)]


class ColStat(C.Structure):
    _fields_ = [("nacols", C.c_int32), ("sweeps", C.c_int32), ("conv", C.c_int32),
                ("nnzw", C.c_int32), ("G", C.c_int64), ("D", C.c_int64),
                ("U", C.c_int64), ("err", C.c_double), ("obj", C.c_double)]


COLSTAT_DTYPE = np.dtype([("nacols", "i4"), ("sweeps", "i4"), ("conv", "i4"),
                          ("nnzw", "i4"), ("G", "i8"), ("D", "i8"), ("U", "i8"),
                          ("err", "f8"), ("obj", "f8")], align=True)


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, "slim_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libslim_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.oracle_learn_cd.restype = C.c_int32
        _lib.oracle_perm_key.restype = C.c_uint32
        _lib.oracle_perm_key.argtypes = [C.c_uint32] * 3
        _lib.oracle_perm_index.restype = C.c_uint32
        _lib.oracle_perm_index.argtypes = [C.c_uint32] * 3
        _lib.oracle_max_threads.restype = C.c_int32
        _lib.oracle_free.argtypes = [C.c_void_p]
        _lib.oracle_set_time_budget.argtypes = [C.c_double]
        _lib.oracle_set_time_budget.restype = None
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _csr_arrays(R, binary=False):
    R = sp.csr_matrix(R)   # (no copy for a CSR input: repeated calls pass the same index array)
    ptr = np.ascontiguousarray(R.indptr, dtype=np.int64)
    ind = np.ascontiguousarray(R.indices, dtype=np.int32)
    val = None if binary else np.ascontiguousarray(R.data, dtype=np.float32)
    return R.shape[0], ptr, ind, val


def learn_cd(R, l1r=1.0, l2r=1.0, optTol=1e-7, maxniters=10000, nthreads=1,
             order=ORDER_GLIBC, seed=1, aty=ATY_FULLSCAN, fp32=False,
             imodel=None, cols=None, binary=False, srand=1, return_stats=False,
             nnbrs=0, simtype=0, chunk=0):
    """Restated SLIM_Learn(algo=cd).  R: scipy CSR (ids used as given; model
    dimension = max id + 1, setup.c:117).  imodel: scipy sparse W of a previous
    solve (warm start through its column view).  Returns W as scipy CSC
    (column iC = regressors of item iC) [+ stats, error, objval]."""
    L = lib()
    nrows, ptr, ind, val = _csr_arrays(R, binary)
    cfg = Cfg(l1r, l2r, optTol, maxniters, nthreads, order, seed, aty, int(fp32), nnbrs, simtype,
              chunk, 0, 0)
    if order == ORDER_GLIBC and srand is not None:
        L.oracle_srand(C.c_uint32(srand))
    ic_ptr = ic_ind = ic_val = None
    ic_n = 0
    if imodel is not None:
        Wc = sp.csc_matrix(imodel)
        Wc.sort_indices()
        ic_ptr = np.ascontiguousarray(Wc.indptr, dtype=np.int64)
        ic_ind = np.ascontiguousarray(Wc.indices, dtype=np.int32)
        ic_val = np.ascontiguousarray(Wc.data, dtype=np.float32)
        ic_n = Wc.shape[1]
    ncols = int(ind.max()) + 1 if ind.size else 0
    stats = np.zeros(ncols, dtype=COLSTAT_DTYPE)
    sel = None if cols is None else np.ascontiguousarray(cols, dtype=np.int32)
    wptr, wind, wval = C.POINTER(C.c_int64)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_float)()
    err, obj = C.c_double(0), C.c_double(0)
    n = L.oracle_learn_cd(C.c_int32(nrows), _p(ptr, C.c_int64), _p(ind, C.c_int32),
                          _p(val, C.c_float), C.byref(cfg),
                          _p(ic_ptr, C.c_int64), _p(ic_ind, C.c_int32),
                          _p(ic_val, C.c_float), C.c_int32(ic_n),
                          C.c_int32(0 if sel is None else sel.size), _p(sel, C.c_int32),
                          C.byref(wptr), C.byref(wind), C.byref(wval),
                          stats.ctypes.data_as(C.POINTER(ColStat)),
                          C.byref(err), C.byref(obj))
    if n < 0:
        raise RuntimeError("oracle_learn_cd failed")
    indptr = np.ctypeslib.as_array(wptr, shape=(n + 1,)).copy()
    nnz = int(indptr[-1])
    indices = np.ctypeslib.as_array(wind, shape=(max(nnz, 1),))[:nnz].copy()
    data = np.ctypeslib.as_array(wval, shape=(max(nnz, 1),))[:nnz].copy()
    for p in (wptr, wind, wval):
        L.oracle_free(C.cast(p, C.c_void_p))
    W = sp.csc_matrix((data, indices, indptr), shape=(n, n))
    if return_stats:
        return W, stats, err.value, obj.value
    return W


def tile_work_order(R, col_begin=0, col_end=None):
    """The engine's work list for a column range: descending Gram work G (sum over the
    column's users of their row lengths), stable (slim_amd/csrc/engine.hip::learn_cd)."""
    R = sp.csr_matrix(R)
    Rc = R.tocsc()
    deg = np.diff(R.indptr).astype(np.int64)
    ncols = int(R.indices.max()) + 1 if R.nnz else 0
    G = np.zeros(ncols, np.int64)
    colof = np.repeat(np.arange(Rc.shape[1]), np.diff(Rc.indptr))
    np.add.at(G, colof, deg[Rc.indices])
    col_end = ncols if col_end is None else col_end
    cols = np.arange(col_begin, col_end)
    return cols[np.argsort(-G[col_begin:col_end], kind="stable")].astype(np.int32)


def learn_cd_tile(R, tileP=32, order=None, l1r=1.0, l2r=1.0, optTol=1e-7, maxniters=10000,
                  nthreads=1, seed=1, binary=False, return_stats=False, tiles=None, nnbrs=0,
                  simtype=0, imodel=None):
    """EstimateModelCD in the tile kernel's visiting order (see oracle_learn_cd_tile).
    tiles=(first, count): walk only those tiles of the work list (their position keys the
    visiting order, so a tile of a larger launch can be checked alone).  imodel: scipy sparse
    W of a previous solve (warm start, estimate.c:453-464 + cd.c:108-110)."""
    L = lib()
    ic_ptr = ic_ind = ic_val = None
    ic_n = 0
    if imodel is not None:
        Wc = sp.csc_matrix(imodel)
        Wc.sort_indices()
        ic_ptr = np.ascontiguousarray(Wc.indptr, dtype=np.int64)
        ic_ind = np.ascontiguousarray(Wc.indices, dtype=np.int32)
        ic_val = np.ascontiguousarray(Wc.data, dtype=np.float32)
        ic_n = Wc.shape[1]
    nrows, ptr, ind, val = _csr_arrays(R, binary)
    if order is None:
        order = tile_work_order(R)
    order = np.ascontiguousarray(order, dtype=np.int32)
    cfg = Cfg(l1r, l2r, optTol, maxniters, nthreads, ORDER_PERM, seed, ATY_GRAM, 0, nnbrs, simtype,
              0, tiles[0] if tiles else 0, tiles[1] if tiles else 0)
    ncols = int(ind.max()) + 1 if ind.size else 0
    stats = np.zeros(ncols, dtype=COLSTAT_DTYPE)
    wptr, wind, wval = C.POINTER(C.c_int64)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_float)()
    err, obj = C.c_double(0), C.c_double(0)
    L.oracle_learn_cd_tile_warm.restype = C.c_int32
    n = L.oracle_learn_cd_tile_warm(C.c_int32(nrows), _p(ptr, C.c_int64), _p(ind, C.c_int32),
                                    _p(val, C.c_float), C.byref(cfg), C.c_int32(tileP),
                                    C.c_int32(order.size), _p(order, C.c_int32),
                                    _p(ic_ptr, C.c_int64), _p(ic_ind, C.c_int32),
                                    _p(ic_val, C.c_float), C.c_int32(ic_n),
                                    C.byref(wptr), C.byref(wind), C.byref(wval),
                                    stats.ctypes.data_as(C.POINTER(ColStat)),
                                    C.byref(err), C.byref(obj))
    if n < 0:
        raise RuntimeError("oracle_learn_cd_tile failed")
    indptr = np.ctypeslib.as_array(wptr, shape=(n + 1,)).copy()
    nnz = int(indptr[-1])
    indices = np.ctypeslib.as_array(wind, shape=(max(nnz, 1),))[:nnz].copy()
    data = np.ctypeslib.as_array(wval, shape=(max(nnz, 1),))[:nnz].copy()
    for p in (wptr, wind, wval):
        L.oracle_free(C.cast(p, C.c_void_p))
    W = sp.csc_matrix((data, indices, indptr), shape=(n, n))
    if return_stats:
        return W, stats, err.value, obj.value
    return W


def learn_admm(R, l1r=1.0, l2r=1.0, nthreads=1, binary=False):
    """EstimateModelADMM restated (estimate.c:38-304).  Returns W as scipy CSR (the model's row
    view: W[i, k] = weight of history item i when scoring candidate k)."""
    L = lib()
    nrows, ptr, ind, val = _csr_arrays(R, binary)
    wptr, wind, wval = C.POINTER(C.c_int64)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_float)()
    L.oracle_learn_admm.restype = C.c_int32
    n = L.oracle_learn_admm(C.c_int32(nrows), _p(ptr, C.c_int64), _p(ind, C.c_int32),
                            _p(val, C.c_float), C.c_double(l1r), C.c_double(l2r),
                            C.c_int32(nthreads), C.byref(wptr), C.byref(wind), C.byref(wval))
    if n < 0:
        raise RuntimeError("oracle_learn_admm failed (%d)" % n)
    indptr = np.ctypeslib.as_array(wptr, shape=(n + 1,)).copy()
    nnz = int(indptr[-1])
    indices = np.ctypeslib.as_array(wind, shape=(max(nnz, 1),))[:nnz].copy()
    data = np.ctypeslib.as_array(wval, shape=(max(nnz, 1),))[:nnz].copy()
    for q in (wptr, wind, wval):
        L.oracle_free(C.cast(q, C.c_void_p))
    return sp.csr_matrix((data, indices, indptr), shape=(n, n))


def _w_rows(W):
    Wr = sp.csr_matrix(W)
    Wr.sort_indices()
    return (Wr.shape[0], np.ascontiguousarray(Wr.indptr, dtype=np.int64),
            np.ascontiguousarray(Wr.indices, dtype=np.int32),
            np.ascontiguousarray(Wr.data, dtype=np.float32))


def predict(W, H, nrcmds=10, binary=False):
    """Py_SLIM_Predict restated: (ids[nusers,nrcmds] filled with -1, scores)."""
    L = lib()
    ncols, wp, wi, wv = _w_rows(W)
    nu, hp, hi, hv = _csr_arrays(H, binary)
    out = np.full(nu * nrcmds, -1, dtype=np.int32)
    sc = np.zeros(nu * nrcmds, dtype=np.float32)
    L.oracle_predict(C.c_int32(ncols), _p(wp, C.c_int64), _p(wi, C.c_int32), _p(wv, C.c_float),
                     C.c_int32(nu), _p(hp, C.c_int64), _p(hi, C.c_int32), _p(hv, C.c_float),
                     C.c_int32(nrcmds), _p(out, C.c_int32), _p(sc, C.c_float))
    return out.reshape(nu, nrcmds), sc.reshape(nu, nrcmds)


def predict_1vsk(W, H, negitems, nrcmds=10, binary=False):
    """Py_SLIM_Predict_1vsk restated: negitems[nusers, nnegs] -> (ids filled with -1, scores)."""
    L = lib()
    ncols, wp, wi, wv = _w_rows(W)
    nu, hp, hi, hv = _csr_arrays(H, binary)
    neg = np.ascontiguousarray(negitems, dtype=np.int32).reshape(nu, -1)
    out = np.full(nu * nrcmds, -1, dtype=np.int32)
    sc = np.zeros(nu * nrcmds, dtype=np.float32)
    L.oracle_predict_1vsk(C.c_int32(ncols), _p(wp, C.c_int64), _p(wi, C.c_int32), _p(wv, C.c_float),
                          C.c_int32(nu), _p(hp, C.c_int64), _p(hi, C.c_int32), _p(hv, C.c_float),
                          C.c_int32(nrcmds), C.c_int32(neg.shape[1]), _p(neg, C.c_int32),
                          _p(out, C.c_int32), _p(sc, C.c_float))
    return out.reshape(nu, nrcmds), sc.reshape(nu, nrcmds)


def evaluate(W, trn, tst, nrcmds=10, binary=False):
    """HR/ARHR per pyapi.c:309-366.  Returns dict(hr, hr_head, hr_tail, arhr, nvalid...)."""
    L = lib()
    ncols, wp, wi, wv = _w_rows(W)
    nu, tp, ti, tv = _csr_arrays(trn, binary)
    nu2, sp_, si, _ = _csr_arrays(tst, True)
    if nu2 < nu:  # test matrix may have fewer trailing rows
        sp_ = np.concatenate([sp_, np.full(nu - nu2, sp_[-1], dtype=np.int64)])
    fm = int(si.max()) + 1 if si.size else 0
    res = np.zeros(4, dtype=np.float64)
    cnt = np.zeros(3, dtype=np.int32)
    L.oracle_eval(C.c_int32(ncols), _p(wp, C.c_int64), _p(wi, C.c_int32), _p(wv, C.c_float),
                  C.c_int32(nu), _p(tp, C.c_int64), _p(ti, C.c_int32), _p(tv, C.c_float),
                  _p(sp_, C.c_int64), _p(si, C.c_int32), C.c_int32(nrcmds), C.c_int32(fm),
                  _p(res, C.c_double), _p(cnt, C.c_int32))
    return dict(hr=res[0], hr_head=res[1], hr_tail=res[2], arhr=res[3],
                nvalid=int(cnt[0]), nvalid_head=int(cnt[1]), nvalid_tail=int(cnt[2]))


def perm_index(p, n, key):
    return int(lib().oracle_perm_index(p, n, key))


def perm_key(seed, item, sweep):
    return int(lib().oracle_perm_key(seed, item, sweep))


def cache_setup(on):
    """Keep the column view (transpose + norms) of the last matrix across calls while the SAME
    index array is passed (bench.py's CPU leg on a 1e9-nnz matrix); cache_setup(False) frees it."""
    lib().oracle_cache_setup(C.c_int32(1 if on else 0))


def set_time_budget(seconds):
    """Wall-clock budget of the estimate phase of the NEXT learn_cd call (bench.py's bounded CPU
    sample): past it no new sweep starts; cut-off columns report conv = -1 (and the D they
    reached), columns never started conv = -2.  Not for anything that is compared."""
    lib().oracle_set_time_budget(float(seconds))


def learn_seconds():
    """Wall time of the estimate phase of the last learn_cd / learn_cd_tile call (the
    reference's LearnTmr), without the transpose and norms of the setup."""
    lib().oracle_learn_seconds.restype = C.c_double
    return float(lib().oracle_learn_seconds())


def physical_cores():
    """(physical cores, hardware threads) of this host, from /proc/cpuinfo."""
    cores = set()
    phys = core = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    n = os.cpu_count() or 1
    return (len(cores) or n), n


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def max_threads():
    return int(lib().oracle_max_threads())


def head_tail(trn, ncols):
    """SLIM_DetermineHeadAndTail restated (api.c:215-245): 0 = head, 1 = tail."""
    nu, tp, ti, _ = _csr_arrays(trn, True)
    fm = np.zeros(ncols, dtype=np.int32)
    lib().oracle_head_tail(C.c_int32(nu), C.c_int32(ncols), _p(tp, C.c_int64), _p(ti, C.c_int32),
                           _p(fm, C.c_int32))
    return fm


def get_topn(W, itemids, ratings=None, nrcmds=10):
    ncols, wp, wi, wv = _w_rows(W)
    ids = np.ascontiguousarray(itemids, dtype=np.int32)
    rt = None if ratings is None else np.ascontiguousarray(ratings, dtype=np.float32)
    rids = np.zeros(nrcmds, np.int32)
    rsc = np.zeros(nrcmds, np.float32)
    lib().oracle_get_topn.restype = C.c_int32
    n = lib().oracle_get_topn(C.c_int32(ncols), _p(wp, C.c_int64), _p(wi, C.c_int32),
                              _p(wv, C.c_float), C.c_int32(ids.size), _p(ids, C.c_int32),
                              _p(rt, C.c_float), C.c_int32(nrcmds), _p(rids, C.c_int32),
                              _p(rsc, C.c_float))
    return rids[:n], rsc[:n]
