"""Host-side mirror of the reference's Python package (``from SLIM import SLIM,
SLIMatrix``) on top of the MI355X engine's libslim.so.

Same class names, method names, argument meaning, defaults, printed messages
and error behaviour as /root/reference/python-package/SLIM/core.py; the code is
new.  Everything numerical happens behind the C ABI (include/slim.h,
include/slim_gpu.h): training on the GPU, top-N scoring in the library.

Reference anchors: parameter defaults and validation core.py:46-242, SLIMatrix
core.py:245-385 (id mapping in first-appearance order :289-351, marshalling
dtypes :353-362), SLIM core.py:388-683.
"""
import ctypes as C
import numbers
import os
import time

import numpy as np
import scipy.sparse as sp

from . import _lib
from .constants import (SLIM_ALGO, SLIM_NOPTIONS, SLIM_OK, SLIM_SIMTYPE, Opt)

try:  # pandas is optional, exactly as in the reference
    from pandas import DataFrame
    PANDAS_INSTALLED = True
except Exception:  # pragma: no cover
    DataFrame = None
    PANDAS_INSTALLED = False


# ---------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------
def _is_int(v):
    return type(v) is int


def _is_real(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


# name -> (default, predicate, complaint)
_PARAMS = (
    ("dbglvl", 0, lambda v: _is_int(v) and v >= 0,
     "Please select dbglvl from {0, 1, 2, 4, 16, 2048}."),
    ("nnbrs", 0, lambda v: _is_int(v) and v >= 0,
     "Please provide non-negative integer value for nnbrs."),
    ("simtype", "cos", lambda v: v in SLIM_SIMTYPE,
     "Please select simtytpe from {'cos', 'jacc', 'dotp'}."),
    ("algo", "cd", lambda v: v in SLIM_ALGO, "Please select algo from {'admm', 'cd'}."),
    ("nthreads", 1, lambda v: _is_int(v) and v > 0,
     "Please provide positive integer value for nthreads."),
    ("niters", 50, lambda v: _is_int(v) and v > 0,
     "Please provide positive integer value for niters."),
    ("nrcmds", 10, lambda v: _is_int(v) and v > 0,
     "Please provide positive integer value for nrcmds."),
    ("l1r", 1.0, lambda v: _is_real(v) and v >= 0, "Please provide non-negative value for l1r."),
    ("l2r", 1.0, lambda v: _is_real(v) and v >= 0, "Please provide non-negative value for l2r."),
    ("optTol", 1e-7, lambda v: _is_real(v) and v >= 0,
     "Please provide non-negative value for optTol."),
)


class _ParamView(object):
    """Uniform get/set/has over a dict or an attribute bag (argparse namespace...)."""

    def __init__(self, params):
        self.p = params
        self.is_dict = isinstance(params, dict)

    def has(self, k):
        return (k in self.p) if self.is_dict else hasattr(self.p, k)

    def get(self, k):
        return self.p[k] if self.is_dict else getattr(self.p, k)

    def set(self, k, v):
        if self.is_dict:
            self.p[k] = v
        else:
            setattr(self.p, k, v)


def check_params(params):
    """Validate ``params`` and fill in the defaults *in place* (the reference
    does the same, core.py:46-198).  Raises TypeError on a bad value."""
    view = _ParamView(params)
    for name, default, good, complaint in _PARAMS:
        if view.has(name):
            if not good(view.get(name)):
                raise TypeError(complaint)
        else:
            view.set(name, default)
    if view.get("nnbrs") > 0 and view.get("algo") != "cd":
        print("A fSLIM model cannot be trained with ADMM. Changing the algorithm to "
              "coordinate descent.")
        view.set("algo", "cd")
    view.set("ordered", 0)  # accepted by the C API, never implemented upstream
    return view


def build_options(params):
    """(ioptions, doptions) arrays for the C API (core.py:200-242)."""
    view = params if isinstance(params, _ParamView) else _ParamView(params)
    iopt = np.full(SLIM_NOPTIONS, -1, dtype=np.int32)
    dopt = np.full(SLIM_NOPTIONS, -1.0, dtype=np.float64)
    iopt[Opt.DBGLVL] = view.get("dbglvl")
    iopt[Opt.NNBRS] = view.get("nnbrs")
    iopt[Opt.SIMTYPE] = SLIM_SIMTYPE[view.get("simtype")]
    iopt[Opt.ALGO] = SLIM_ALGO[view.get("algo")]
    iopt[Opt.NTHREADS] = view.get("nthreads")
    iopt[Opt.ORDERED] = view.get("ordered")
    iopt[Opt.MAXNITERS] = view.get("niters")
    iopt[Opt.NRCMDS] = view.get("nrcmds")
    dopt[Opt.L1R] = view.get("l1r")
    dopt[Opt.L2R] = view.get("l2r")
    dopt[Opt.OPTTOL] = view.get("optTol")
    # engine extensions ride along when present (include/slim_gpu.h)
    for key, slot in (("gpu_seed", Opt.GPU_SEED), ("gpu_device", Opt.GPU_DEVICE),
                      ("gpu_kernel", Opt.GPU_KERNEL), ("gpu_colbegin", Opt.GPU_COLBEGIN),
                      ("gpu_colend", Opt.GPU_COLEND)):
        if view.has(key):
            iopt[slot] = int(view.get(key))
    return iopt, dopt


def _prepare(params):
    if not isinstance(params, dict) and not hasattr(params, "__dict__"):
        raise TypeError("Parameter type %s is not supported!" % type(params).__name__)
    return check_params(params)


# ---------------------------------------------------------------------------
# SLIMatrix
# ---------------------------------------------------------------------------
def _enumerate_first_seen(keys, table=None, names=None):
    """Dense ids in order of first appearance (dict insertion order)."""
    table = {} if table is None else table
    names = [] if names is None else names
    for k in keys:
        if k not in table:
            table[k] = len(names)
            names.append(k)
    return table, names


class SLIMatrix(object):
    """Training / history matrix handed to :class:`SLIM`.

    ``data``: scipy CSR (ids = positions) or user-item-rating triplets as a
    list of lists, a 2-d numpy array or a pandas DataFrame (raw ids are mapped
    to dense ids in first-appearance order).  ``oldmat``: a SLIMatrix or SLIM
    whose id maps should be reused (events outside them are dropped)."""

    def __init__(self, data, oldmat=None):
        self._lib = _lib.load()
        self.handle = None
        if sp.isspmatrix_csr(data):
            self.nUsers, self.nItems = data.shape
            if isinstance(oldmat, SLIMatrix) and (self.nUsers != oldmat.nUsers or
                                                  self.nItems != oldmat.nItems):
                raise TypeError("The size of the input matrix does not match the size of oldmat.")
            if isinstance(oldmat, SLIM) and self.nItems != oldmat.id2item.size:
                raise TypeError("The size of the input matrix does not match the size of oldmat.")
            self.id2item = np.arange(self.nItems)
            self.item2id = self.id2item
            self.id2user = np.arange(self.nUsers)
            self.user2id = self.id2user
            self._set_csr(data)
        elif isinstance(data, (list, np.ndarray)):
            self.data_from_np2d(data, oldmat)
        elif PANDAS_INSTALLED and isinstance(data, DataFrame):
            self.data_from_np2d(data.values, oldmat)
        else:
            raise TypeError(
                "Input data type %s is not supported. Please provide ijv triplets in "
                "numpy.ndarray/list[List]/pandas.DataFrame or a row based sparse matrix in "
                "scipy csr_matrix." % type(data).__name__)

    def __del__(self):
        try:
            if self.handle is not None:
                self._lib.Py_csr_free(self.handle)
                self.handle = None
        except Exception:
            pass

    def data_from_np2d(self, data, oldmat=None):
        if oldmat is not None:
            assert isinstance(oldmat, (SLIMatrix, SLIM)), \
                "Please feed in a SLIMatrix object or a SLIM model for oldmat."
            self.id2item = oldmat.id2item.copy()
            self.item2id = oldmat.item2id.copy()
            if isinstance(oldmat, SLIMatrix):
                self.id2user = oldmat.id2user.copy()
                self.user2id = oldmat.user2id.copy()
            else:  # a model knows items only: users are enumerated afresh
                self.user2id, self.id2user = _enumerate_first_seen(t[0] for t in data)
        else:
            self.user2id, users = _enumerate_first_seen(t[0] for t in data)
            self.item2id, items = _enumerate_first_seen(t[1] for t in data)
            self.id2user = np.array(users)
            self.id2item = np.array(items)

        rows, cols, vals, missed = [], [], [], 0
        u2i, i2i = self.user2id, self.item2id
        for t in data:
            if t[0] in u2i and t[1] in i2i:
                rows.append(u2i[t[0]])
                cols.append(i2i[t[1]])
                vals.append(t[2])
            else:
                missed += 1
        if missed:
            print("%d of the events fall out of the range of oldmat. Partial entries collected."
                  % missed)
        self.nUsers = len(self.id2user)
        self.nItems = len(self.id2item)
        self._set_csr(sp.csr_matrix((vals, (rows, cols)), shape=(self.nUsers, self.nItems)))

    def _set_csr(self, R):
        handle = C.c_void_p()
        indptr = np.ascontiguousarray(R.indptr, dtype=np.intp)
        indices = np.ascontiguousarray(R.indices, dtype=np.int32)
        values = np.ascontiguousarray(R.data, dtype=np.float32)
        rc = self._lib.Py_csr_wrapper(R.shape[0], indptr, indices,
                                      values.ctypes.data_as(C.c_void_p), C.byref(handle))
        if rc != SLIM_OK:
            raise RuntimeError("Py_csr_wrapper failed (%d): %s" % (rc, _lib.last_error()))
        self.handle = handle


# ---------------------------------------------------------------------------
# SLIM
# ---------------------------------------------------------------------------
class SLIM(object):
    """Item-item sparse linear model; ``train`` runs on the GPU engine."""

    def __init__(self):
        self._lib = _lib.load()
        self.ismodel = 0
        self.handle = None

    def __del__(self):
        try:
            if self.handle is not None:
                self._lib.Py_csr_free(self.handle)
                self.handle = None
        except Exception:
            pass

    # -- training ------------------------------------------------------------
    def train(self, params, data):
        assert type(data) == SLIMatrix, "trndata must be a SLIMatrix object."
        self.nItems = data.nItems
        iopt, dopt = build_options(_prepare(params))
        handle = C.c_void_p()
        t0 = time.time()
        rc = self._lib.Py_SLIM_Learn(data.handle, iopt, dopt, C.byref(handle))
        elapsed = time.time() - t0
        if self.handle is not None:
            self._lib.Py_csr_free(self.handle)
        self.ismodel = rc
        self.handle = handle if rc == SLIM_OK else None
        self.id2item = data.id2item.copy()
        self.item2id = data.item2id.copy()
        if rc != SLIM_OK:
            raise RuntimeError("Something went wrong with model estimation. [%s]"
                               % _lib.last_error())
        print("Learning takes %.3f secs." % elapsed)

    def mselect(self, params, trndata, tstdata, arrayl1, arrayl2, nrcmds):
        assert type(trndata) == SLIMatrix, "trndata must be a SLIMatrix object."
        assert type(tstdata) == SLIMatrix, "tstdata must be a SLIMatrix object."
        assert type(arrayl1) in [list, np.ndarray], "Please provide a list of l1 values."
        assert type(arrayl2) in [list, np.ndarray], "Please provide a list of l2 values."
        view = _prepare(params)
        view.set("nrcmds", nrcmds)
        iopt, dopt = build_options(view)
        if len(arrayl1) < 1:
            raise TypeError("The l1 array must not be empty.")
        if len(arrayl2) < 1:
            raise TypeError("The l2 array must not be empty.")
        best = [C.c_double(0.0) for _ in range(8)]
        t0 = time.time()
        rc = self._lib.Py_SLIM_Mselect(
            trndata.handle, tstdata.handle, iopt, dopt,
            np.ascontiguousarray(np.sort(arrayl1), dtype=np.float64),
            np.ascontiguousarray(np.sort(arrayl2), dtype=np.float64),
            len(arrayl1), len(arrayl2), *[C.byref(b) for b in best])
        elapsed = time.time() - t0
        l1hr, l2hr, hrhr, arhr, l1ar, l2ar, hrar, arar = [b.value for b in best]
        if rc != SLIM_OK:
            raise RuntimeError(
                "Something went wrong with model estimation or evaluation when l1=%.4f, "
                "l2=%.4f. Please check the input matrix. [%s]" % (l1hr, l2hr, _lib.last_error()))
        print("Model selection takes %.3f secs." % elapsed)
        print("The best HR is achieved by, l1: %.4f, l2:%.4f, HR:%.4f, AR:%.4f."
              % (l1hr, l2hr, hrhr, arhr))
        print("The best AR is achieved by, l1: %.4f, l2:%.4f, HR:%.4f, AR:%.4f."
              % (l1ar, l2ar, hrar, arar))
        self.mselect_result = dict(bestHR=(l1hr, l2hr, hrhr, arhr), bestAR=(l1ar, l2ar, hrar, arar))

    # -- prediction ----------------------------------------------------------
    def predict(self, data, nrcmds=10, outfile=None, negitems=None, nnegs=0, returnscores=False):
        if self.ismodel != SLIM_OK:
            raise TypeError("Model not found. Please train a model.")
        assert self.nItems == data.nItems, \
            "The shape of the input matrix should match the model."
        res = np.full(data.nUsers * nrcmds, -1, dtype=np.int32)
        scores = np.zeros(data.nUsers * nrcmds, dtype=np.float32)
        user_is_dict = isinstance(data.user2id, dict)

        if negitems is not None:
            assert nnegs >= nrcmds, ("The number of negative items must be larger than the "
                                     "number of items to be recommended.")
            if user_is_dict:
                assert data.user2id.keys() == negitems.keys(), \
                    "The users in the negative items should be the same with the input matrix."
            else:
                assert np.array_equal(data.user2id, np.array(sorted(negitems.keys()))), \
                    "The users in the negative items should be the same with the input matrix."
            cand = np.full(data.nUsers * nnegs, -1, dtype=np.int32)
            unknown = 0
            for user, items in negitems.items():
                assert len(items) == nnegs, "The number of negative items should match nngs."
                base = data.user2id[user] * nnegs
                for j, it in enumerate(items):
                    try:
                        cand[base + j] = self.item2id[it]
                    except (KeyError, IndexError):
                        unknown += 1
            if unknown:
                print("%d negative items not in the training set." % unknown)
            rc = self._lib.Py_SLIM_Predict_1vsk(nrcmds, nnegs, self.handle, data.handle, cand,
                                                res, scores)
        else:
            rc = self._lib.Py_SLIM_Predict(nrcmds, self.handle, data.handle, res, scores)
        if rc != SLIM_OK:
            raise RuntimeError(
                "Something went wrong during prediction. Please check 1) if the model is "
                "estimated correctly; 2) if the input matrix for prediction is correct.")

        # unfilled slots hold -1 and, as in the reference (core.py:584), index the
        # item map from the end
        res = np.asarray(self.id2item)[res].reshape(data.nUsers, nrcmds)
        scores = scores.reshape(data.nUsers, nrcmds)
        pairs = data.user2id.items() if user_is_dict else ((k, k) for k in data.user2id)
        out, outscores = {}, {}
        for key, row in pairs:
            out[key] = res[row, :]
            outscores[key] = scores[row, :]
        if outfile:
            with open(outfile, "w") as f:
                for key, value in out.items():
                    f.write(str(key) + ": " + np.array2string(value, max_line_width=np.inf) + "\n")
                    if returnscores:
                        f.write(str(key) + ": " +
                                np.array2string(outscores[key], max_line_width=np.inf) + "\n")
        return (out, outscores) if returnscores else out

    # -- persistence -----------------------------------------------------------
    def save_model(self, modelfname, mapfname):
        if self.ismodel != SLIM_OK:
            raise RuntimeError("Not exist a model to save.")
        self._lib.Py_csr_save(self.handle, modelfname.encode("utf-8"))
        np.savetxt(mapfname, self.id2item, fmt="%s")

    def load_model(self, modelfname, mapfname):
        if not (os.path.isfile(modelfname) and os.path.isfile(mapfname)):
            raise RuntimeError("File does not exist or invalid filename.")
        if self.ismodel == SLIM_OK and self.handle is not None:
            self._lib.Py_csr_free(self.handle)
        handle = C.c_void_p()
        self.ismodel = self._lib.Py_csr_load(C.byref(handle), modelfname.encode("utf-8"))
        self.handle = handle if self.ismodel == SLIM_OK else None
        try:
            self.id2item = np.genfromtxt(mapfname, dtype=np.int32)
        except Exception:
            self.id2item = np.genfromtxt(mapfname)
        self.id2item = np.atleast_1d(self.id2item)
        self.item2id = {self.id2item[i]: i for i in range(len(self.id2item))}
        self.nItems = len(self.id2item)
        if self.ismodel != SLIM_OK:
            raise RuntimeError("Fail to laod the model.")

    def to_csr(self, returnmap=False):
        if self.ismodel != SLIM_OK:
            raise RuntimeError("Not exist a model to export.")
        nnz = C.c_int(0)
        self._lib.Py_csr_stat(self.handle, C.byref(nnz))
        view = C.cast(self.handle, C.POINTER(_lib.CsrView)).contents
        nrows = max(int(view.nrows), 0)
        indptr = np.zeros(max(self.nItems, nrows) + 1, dtype=np.int32)
        indices = np.zeros(nnz.value, dtype=np.int32)
        data = np.ones(nnz.value, dtype=np.float32)
        self._lib.Py_csr_export(self.handle, indptr, indices, data)
        indptr[nrows + 1:] = indptr[nrows]  # trailing unrated items: empty rows
        model = sp.csr_matrix((data, indices, indptr[:self.nItems + 1]),
                              shape=(self.nItems, self.nItems))
        return (model, self.id2item[:]) if returnmap else model
