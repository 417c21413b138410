"""Python handle on the engine extensions of libslim.so (SLIMGPU_*, include/slim_gpu.h):
a training matrix staged in HBM and repeated / column-sharded CD solves on it.

Used by bench.py, the multi-GPU helper (slim_amd/distributed.py) and the parity
tests.  Device buffers are passed as raw pointers (``tensor.data_ptr()``); torch is
plumbing only and is not imported here.
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib
from .constants import SLIM_NOPTIONS, SLIM_OK, Opt

KERNEL_AUTO, KERNEL_WAVE_LDS, KERNEL_WAVE_HBM, KERNEL_TILE, KERNEL_TILE16, KERNEL_GRAM = 0, 1, 2, 3, 4, 5


def make_options(l1r=1.0, l2r=1.0, optTol=1e-7, niters=10000, seed=1, col_begin=None,
                 col_end=None, kernel=KERNEL_AUTO, device=None, dbglvl=0, cluster=None, nnbrs=0,
                 simtype=0, heavy_tiles=None, heavy_cluster=None, ngpus=None, shard=None):
    iopt = np.full(SLIM_NOPTIONS, -1, dtype=np.int32)
    dopt = np.full(SLIM_NOPTIONS, -1.0, dtype=np.float64)
    iopt[Opt.DBGLVL] = dbglvl
    iopt[Opt.MAXNITERS] = niters
    iopt[Opt.GPU_SEED] = seed
    iopt[Opt.GPU_KERNEL] = kernel
    iopt[Opt.NNBRS] = nnbrs
    iopt[Opt.SIMTYPE] = simtype
    if col_begin is not None:
        iopt[Opt.GPU_COLBEGIN] = col_begin
    if col_end is not None:
        iopt[Opt.GPU_COLEND] = col_end
    if device is not None:
        iopt[Opt.GPU_DEVICE] = device
    if cluster is not None:
        iopt[Opt.GPU_CLUSTER] = cluster
    if heavy_tiles is not None:
        iopt[Opt.GPU_HEAVYTILES] = heavy_tiles
    if heavy_cluster is not None:
        iopt[Opt.GPU_HEAVYCLUSTER] = heavy_cluster
    if ngpus is not None:
        iopt[Opt.GPU_NGPUS] = ngpus
    if shard is not None:  # (index, count)
        iopt[Opt.GPU_SHARDINDEX], iopt[Opt.GPU_SHARDCOUNT] = shard
    dopt[Opt.L1R], dopt[Opt.L2R], dopt[Opt.OPTTOL] = l1r, l2r, optTol
    return iopt, dopt


def model_to_scipy(lib, handle, free=True):
    """Copy a model handle (slim_csr_t) into a scipy CSC matrix (column iC =
    regressors of item iC) and optionally release the handle."""
    view = C.cast(handle, C.POINTER(_lib.CsrView)).contents
    n = int(view.ncols)
    colptr = np.ctypeslib.as_array(view.colptr, shape=(n + 1,)).astype(np.int64)
    nnz = int(colptr[-1])
    if nnz:
        colind = np.ctypeslib.as_array(view.colind, shape=(nnz,)).copy()
        colval = np.ctypeslib.as_array(view.colval, shape=(nnz,)).copy()
    else:
        colind = np.zeros(0, np.int32)
        colval = np.zeros(0, np.float32)
    W = sp.csc_matrix((colval, colind, colptr), shape=(n, n))
    if free:
        h = C.c_void_p(handle if isinstance(handle, int) else handle.value)
        lib.SLIM_FreeModel(C.byref(h))
    return W


class ColumnStats(object):
    def __init__(self, lib, ncols):
        self.nacols = np.zeros(ncols, np.int32)
        self.sweeps = np.zeros(ncols, np.int32)
        self.conv = np.zeros(ncols, np.int32)
        self.G = np.zeros(ncols, np.int64)
        self.D = np.zeros(ncols, np.int64)
        self.U = np.zeros(ncols, np.int64)
        rc = lib.SLIMGPU_LastColumnStats(ncols, *[a.ctypes.data_as(C.c_void_p) for a in
                                                  (self.nacols, self.sweeps, self.conv, self.G,
                                                   self.D, self.U)])
        if rc != SLIM_OK:
            raise RuntimeError("SLIMGPU_LastColumnStats failed (%d)" % rc)


class ResidentModel(object):
    """A learned model that stays in HBM (SLIMGPU_LearnResident): both views on the device, usable as
    the next solve's warm start without an upload; `fetch()` forms SLIM_Learn's host model."""

    def __init__(self, lib, handle):
        self._lib = lib
        self.handle = C.c_void_p(handle)

    @property
    def nnz(self):
        return int(self._lib.SLIMGPU_ModelNnz(self.handle))

    def fetch_begin(self):
        """Start the copy to the host on its own stream + host thread (runs beside the next solve)."""
        st = self._lib.SLIMGPU_ModelFetchBegin(self.handle)
        if st != SLIM_OK:
            raise RuntimeError("SLIMGPU_ModelFetchBegin failed (%d): %s" % (st, _lib.last_error()))

    def fetch(self, return_handle=False):
        """The host model (joins a begun fetch): scipy CSC, or the slim_t handle (SLIM_FreeModel)."""
        st = C.c_int32(0)
        h = self._lib.SLIMGPU_ModelFetch(self.handle, C.byref(st))
        if not h:
            raise RuntimeError("SLIMGPU_ModelFetch failed (%d): %s" % (st.value, _lib.last_error()))
        return h if return_handle else model_to_scipy(self._lib, h)

    def free(self):
        if self.handle:
            self._lib.SLIMGPU_ModelFree(C.byref(self.handle))
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass


class DeviceMatrix(object):
    """Training matrix resident in HBM (CSR + column view + norms)."""

    def __init__(self, handle, keepalive=None):
        self._lib = _lib.load()
        self.handle = C.c_void_p(handle)
        self._keep = keepalive
        nr, nc, nz = C.c_int32(), C.c_int32(), C.c_int64()
        self._lib.SLIMGPU_MatrixInfo(self.handle, C.byref(nr), C.byref(nc), C.byref(nz))
        self.nrows, self.ncols, self.nnz = nr.value, nc.value, nz.value
        self.device = int(self._lib.SLIMGPU_MatrixDevice(self.handle))   # the HIP device of its buffers

    @classmethod
    def from_scipy(cls, R, binary=False, device=None):
        lib = _lib.load()
        R = sp.csr_matrix(R)
        ptr = np.ascontiguousarray(R.indptr, dtype=np.intp)
        ind = np.ascontiguousarray(R.indices, dtype=np.int32)
        val = None if binary else np.ascontiguousarray(R.data, dtype=np.float32)
        iopt, _ = make_options(device=device)
        st = C.c_int32(0)
        h = lib.SLIMGPU_MatrixFromHost(R.shape[0], ptr, ind,
                                       None if val is None else val.ctypes.data_as(C.c_void_p),
                                       iopt.ctypes.data_as(C.c_void_p), C.byref(st))
        if not h:
            raise RuntimeError("SLIMGPU_MatrixFromHost failed (%d): %s" % (st.value, _lib.last_error()))
        return cls(h)

    @classmethod
    def from_device_ptrs(cls, nrows, ncols, rowptr_ptr, rowind_ptr, rowval_ptr, keepalive=None,
                         device=None):
        """Adopt int64 rowptr / int32 rowind / float32 rowval (or 0) already in HBM."""
        lib = _lib.load()
        iopt, _ = make_options(device=device)
        st = C.c_int32(0)
        h = lib.SLIMGPU_MatrixFromDevice(nrows, ncols, C.c_void_p(rowptr_ptr),
                                         C.c_void_p(rowind_ptr),
                                         C.c_void_p(rowval_ptr) if rowval_ptr else None,
                                         iopt.ctypes.data_as(C.c_void_p), C.byref(st))
        if not h:
            raise RuntimeError("SLIMGPU_MatrixFromDevice failed (%d): %s" % (st.value, _lib.last_error()))
        return cls(h, keepalive)

    def close(self):
        if self.handle is not None and self.handle.value:
            self._lib.SLIMGPU_MatrixFree(C.byref(self.handle))
        self.handle = None
        self._keep = None

    __del__ = close

    def column_view(self):
        colptr = np.zeros(self.ncols + 1, np.int64)
        colind = np.zeros(max(self.nnz, 1), np.int32)
        colval = np.zeros(max(self.nnz, 1), np.float32)
        cnorm = np.zeros(self.ncols, np.float32)
        rc = self._lib.SLIMGPU_MatrixGetColumnView(
            self.handle, *[a.ctypes.data_as(C.c_void_p) for a in (colptr, colind, colval, cnorm)])
        if rc != SLIM_OK:
            raise RuntimeError("SLIMGPU_MatrixGetColumnView failed: %s" % _lib.last_error())
        return colptr, colind[:self.nnz], colval[:self.nnz], cnorm

    def column_cost(self):
        cost = np.zeros(self.ncols, np.int64)
        rc = self._lib.SLIMGPU_MatrixColumnCost(self.handle, cost.ctypes.data_as(C.c_void_p))
        if rc != SLIM_OK:
            raise RuntimeError("SLIMGPU_MatrixColumnCost failed")
        return cost

    def learn(self, imodel=None, return_handle=False, columns=None, **opts):
        """SLIMGPU_Learn (or SLIMGPU_LearnColumns when an explicit list of item columns is
        given).  Returns (W as scipy CSC, stats dict)."""
        iopt, dopt = make_options(**opts)
        st = C.c_int32(0)
        ih = None
        tmp = None
        if imodel is not None:
            if isinstance(imodel, (int, C.c_void_p)):
                ih = imodel
            else:  # scipy matrix -> temporary handle with a column view
                tmp = _scipy_to_model_handle(self._lib, imodel)
                ih = tmp
        if columns is not None:
            cols = np.ascontiguousarray(columns, dtype=np.int32)
            h = self._lib.SLIMGPU_LearnColumns(self.handle, cols.size, cols,
                                               iopt.ctypes.data_as(C.c_void_p),
                                               dopt.ctypes.data_as(C.c_void_p), ih, C.byref(st))
        else:
            h = self._lib.SLIMGPU_Learn(self.handle, iopt.ctypes.data_as(C.c_void_p),
                                        dopt.ctypes.data_as(C.c_void_p), ih, C.byref(st))
        if tmp is not None:
            self._lib.SLIM_FreeModel(C.byref(tmp))
        if not h:
            raise RuntimeError("SLIMGPU_Learn failed (%d): %s" % (st.value, _lib.last_error()))
        stats = _lib.Stats()
        self._lib.SLIMGPU_LastStats(C.byref(stats))
        if return_handle:
            return h, stats.as_dict()
        return model_to_scipy(self._lib, h), stats.as_dict()

    def learn_resident(self, warm=None, **opts):
        """SLIMGPU_LearnResident: like learn(), but the model stays in HBM (ResidentModel); `warm`
        is a ResidentModel of an earlier solve (no upload).  Returns (ResidentModel, stats dict)."""
        iopt, dopt = make_options(**opts)
        st = C.c_int32(0)
        h = self._lib.SLIMGPU_LearnResident(self.handle, iopt.ctypes.data_as(C.c_void_p),
                                            dopt.ctypes.data_as(C.c_void_p),
                                            warm.handle if warm is not None else None, C.byref(st))
        if not h:
            raise RuntimeError("SLIMGPU_LearnResident failed (%d): %s" % (st.value, _lib.last_error()))
        stats = _lib.Stats()
        self._lib.SLIMGPU_LastStats(C.byref(stats))
        return ResidentModel(self._lib, h), stats.as_dict()

    def column_stats(self):
        return ColumnStats(self._lib, self.ncols)

    def expect_solves(self, n):
        """Announce n solves of this matrix (a grid): KERNEL_AUTO may then build G = R^T R once
        and solve in item space (SLIMGPU_MatrixExpectSolves)."""
        self._lib.SLIMGPU_MatrixExpectSolves(self.handle, int(n))

    # -- G = R^T R of item-space CD in row blocks (slim_gpu.h; slim_amd.distributed.build_gram_sharded)
    def gram_build_rows(self, row_begin, row_end):
        """Form rows [row_begin, row_end) of G on this handle (every entry of each)."""
        st = self._lib.SLIMGPU_MatrixGramBuildRows(self.handle, int(row_begin), int(row_end))
        if st != SLIM_OK:
            raise RuntimeError("SLIMGPU_MatrixGramBuildRows failed (%d): %s" % (st, _lib.last_error()))

    def gram_view(self):
        """(device pointer, floats per row, rows) of the handle's G."""
        p, ld, n = C.c_void_p(), C.c_int64(), C.c_int32()
        st = self._lib.SLIMGPU_MatrixGramView(self.handle, C.byref(p), C.byref(ld), C.byref(n))
        if st != SLIM_OK:
            raise RuntimeError("SLIMGPU_MatrixGramView failed (%d): %s" % (st, _lib.last_error()))
        return p.value, ld.value, n.value

    def gram_rows_tensor(self, row_begin, row_end):
        """Rows [row_begin, row_end) of G as a torch tensor that ALIASES the engine's buffer
        (a contiguous (rows, ld) float32 block on this handle's device)."""
        import torch
        ptr, ld, n = self.gram_view()
        if not 0 <= row_begin <= row_end <= n:
            raise ValueError("gram_rows_tensor: rows [%d, %d) outside [0, %d)" % (row_begin, row_end, n))

        class _Alias:
            pass
        a = _Alias()
        a.__cuda_array_interface__ = {
            "shape": (int(row_end - row_begin), int(ld)), "typestr": "<f4",
            "data": (int(ptr) + 4 * int(ld) * int(row_begin), False), "version": 2, "strides": None}
        # (on the handle's device, not the current one: a copy instead of an alias would swallow
        # the broadcast of build_gram_sharded and commit a G with missing rows)
        t = torch.as_tensor(a, device=torch.device("cuda", self.device))
        if t.numel() and t.data_ptr() != a.__cuda_array_interface__["data"][0]:
            raise RuntimeError("gram_rows_tensor: torch copied the block instead of aliasing it")
        return t

    def gram_commit(self):
        """Every row of G is in place: item-space solves may use it (byte planes are formed)."""
        st = self._lib.SLIMGPU_MatrixGramCommit(self.handle)
        if st != SLIM_OK:
            raise RuntimeError("SLIMGPU_MatrixGramCommit failed (%d): %s" % (st, _lib.last_error()))


def _scipy_to_model_handle(lib, W):
    """Model handle (row + column views) from a scipy matrix, via the text-free
    route: Py_csr_wrapper on W's rows, then a binary round trip adds columns."""
    import os
    import tempfile
    Wr = sp.csr_matrix(W)
    Wr.sort_indices()
    h = C.c_void_p()
    ptr = np.ascontiguousarray(Wr.indptr, dtype=np.intp)
    ind = np.ascontiguousarray(Wr.indices, dtype=np.int32)
    val = np.ascontiguousarray(Wr.data, dtype=np.float32)
    if ind.size == 0 or ind.max() + 1 < Wr.shape[1]:
        pass  # ncols of the handle = max id + 1; fixed up by the binary round trip below
    lib.Py_csr_wrapper(Wr.shape[0], ptr, ind, val.ctypes.data_as(C.c_void_p), C.byref(h))
    view = C.cast(h, C.POINTER(_lib.CsrView)).contents
    view.ncols = Wr.shape[1]
    fd, path = tempfile.mkstemp(suffix=".slimbin")
    os.close(fd)
    try:
        lib.SLIM_WriteModel(h, path.encode())
        out = C.c_void_p(lib.SLIM_ReadModel(path.encode()))
    finally:
        os.unlink(path)
        lib.Py_csr_free(h)
    return out
