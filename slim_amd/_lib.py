"""Loader for the in-tree C-ABI library slim_amd/libslim.so.

The library is the product: there is no Python or CPU fallback for training.
If it is missing, or if a call reports an error, the caller gets an exception
that says so.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLIM_AMD_LIB points at another build of the same library (A/B comparisons of kernels)
LIB_PATH = os.environ.get("SLIM_AMD_LIB") or os.path.join(_HERE, "libslim.so")

f64_1d = np.ctypeslib.ndpointer(dtype=np.float64, ndim=1, flags="C_CONTIGUOUS")
f32_1d = np.ctypeslib.ndpointer(dtype=np.float32, ndim=1, flags="C_CONTIGUOUS")
i32_1d = np.ctypeslib.ndpointer(dtype=np.int32, ndim=1, flags="C_CONTIGUOUS")
i64_1d = np.ctypeslib.ndpointer(dtype=np.int64, ndim=1, flags="C_CONTIGUOUS")
isz_1d = np.ctypeslib.ndpointer(dtype=np.intp, ndim=1, flags="C_CONTIGUOUS")


class Stats(C.Structure):
    """slimgpu_stats_t (include/slim_gpu.h)."""
    _fields_ = [("ncols_solved", C.c_int32), ("kernel", C.c_int32), ("nwaves", C.c_int32),
                ("lds_bytes", C.c_int32), ("setup_ms", C.c_double), ("kernel_ms", C.c_double),
                ("gather_ms", C.c_double), ("total_ms", C.c_double),
                ("G", C.c_int64), ("D", C.c_int64), ("U", C.c_int64), ("nnzW", C.c_int64),
                ("sweeps", C.c_int64), ("visits", C.c_int64), ("alg_bytes", C.c_double),
                ("error", C.c_double), ("objval", C.c_double), ("gram_build_ms", C.c_double),
                ("gram_rows", C.c_int64), ("gram_bytes", C.c_double),
                ("gram_alloc_ms", C.c_double), ("gram_sums_ms", C.c_double),
                ("gram_sums_kernel_ms", C.c_double), ("gram_pack_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class CsrView(C.Structure):
    """slim_csr_t (include/slim_gpu.h): the object behind every handle."""
    _fields_ = ([("nrows", C.c_int32), ("ncols", C.c_int32),
                 ("rowptr", C.POINTER(C.c_ssize_t)), ("colptr", C.POINTER(C.c_ssize_t)),
                 ("rowind", C.POINTER(C.c_int32)), ("colind", C.POINTER(C.c_int32))] +
                [(n, C.POINTER(C.c_int32)) for n in
                 ("rowids", "colids", "rlabels", "clabels", "rmap", "cmap")] +
                [(n, C.POINTER(C.c_float)) for n in
                 ("rowval", "colval", "rnorms", "cnorms", "rsums", "csums", "rsizes",
                  "csizes", "rvols", "cvols", "rwgts", "cwgts")])


_SIGNATURES = {
    # slim.h
    "SLIM_iSetDefaults": (C.c_int32, [i32_1d]),
    "SLIM_dSetDefaults": (C.c_int32, [f64_1d]),
    "SLIM_Learn": (C.c_void_p, [C.c_int32, isz_1d, i32_1d, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.POINTER(C.c_int32)]),
    "SLIM_GetTopN": (C.c_int32, [C.c_void_p, C.c_int32, i32_1d, C.c_void_p, C.c_void_p,
                                 C.c_int32, i32_1d, f32_1d]),
    "SLIM_WriteModel": (C.c_int32, [C.c_void_p, C.c_char_p]),
    "SLIM_ReadModel": (C.c_void_p, [C.c_char_p]),
    "SLIM_FreeModel": (None, [C.POINTER(C.c_void_p)]),
    "SLIM_DetermineHeadAndTail": (C.POINTER(C.c_int32), [C.c_int32, C.c_int32, isz_1d, i32_1d]),
    # Py_* (slim_gpu.h section 2)
    "Py_csr_wrapper": (C.c_int32, [C.c_int32, isz_1d, i32_1d, C.c_void_p, C.c_void_p]),
    "Py_csr_save": (C.c_int32, [C.c_void_p, C.c_char_p]),
    "Py_csr_load": (C.c_int32, [C.c_void_p, C.c_char_p]),
    "Py_csr_free": (C.c_int32, [C.c_void_p]),
    "Py_csr_stat": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "Py_csr_export": (C.c_int32, [C.c_void_p, i32_1d, i32_1d, f32_1d]),
    "Py_SLIM_Learn": (C.c_int32, [C.c_void_p, i32_1d, f64_1d, C.c_void_p]),
    "Py_SLIM_Mselect": (C.c_int32, [C.c_void_p, C.c_void_p, i32_1d, f64_1d, f64_1d, f64_1d,
                                    C.c_int32, C.c_int32] + [C.c_void_p] * 8),
    "Py_SLIM_GetTopN": (C.c_int32, [C.c_void_p, C.c_int32, i32_1d, C.c_void_p, C.c_int32,
                                    i32_1d, f32_1d, C.c_int32]),
    "Py_SLIM_GetTopN_1vsk": (C.c_int32, [C.c_void_p, C.c_int32, i32_1d, C.c_void_p, C.c_int32,
                                         i32_1d, f32_1d, C.c_int32, i32_1d, C.c_int32]),
    "Py_SLIM_Predict": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, i32_1d, f32_1d]),
    "Py_SLIM_Predict_1vsk": (C.c_int32, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, i32_1d,
                                         i32_1d, f32_1d]),
    # SLIMGPU_* (slim_gpu.h section 3)
    "SLIMGPU_MatrixFromHost": (C.c_void_p, [C.c_int32, isz_1d, i32_1d, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_int32)]),
    "SLIMGPU_MatrixFromDevice": (C.c_void_p, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "SLIMGPU_MatrixFree": (None, [C.POINTER(C.c_void_p)]),
    "SLIMGPU_MatrixInfo": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int64)]),
    "SLIMGPU_MatrixGetColumnView": (C.c_int32, [C.c_void_p] * 5),
    "SLIMGPU_MatrixColumnCost": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "SLIMGPU_MatrixExpectSolves": (None, [C.c_void_p, C.c_int32]),
    "SLIMGPU_MatrixDevice": (C.c_int32, [C.c_void_p]),
    "SLIMGPU_MatrixGramBuildRows": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "SLIMGPU_MatrixGramView": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int32)]),
    "SLIMGPU_MatrixGramCommit": (C.c_int32, [C.c_void_p]),
    "SLIMGPU_Learn": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_int32)]),
    "SLIMGPU_LearnResident": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_int32)]),
    "SLIMGPU_ModelNnz": (C.c_int64, [C.c_void_p]),
    "SLIMGPU_ModelFetchBegin": (C.c_int32, [C.c_void_p]),
    "SLIMGPU_ModelFetch": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_int32)]),
    "SLIMGPU_ModelFree": (None, [C.POINTER(C.c_void_p)]),
    "SLIMGPU_ModelPredict": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "SLIMGPU_LearnColumns": (C.c_void_p, [C.c_void_p, C.c_int32, i32_1d, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.POINTER(C.c_int32)]),
    "SLIMGPU_Predict": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, i32_1d, f32_1d]),
    "SLIMGPU_Predict1vsK": (C.c_int32, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, i32_1d,
                                        i32_1d, f32_1d]),
    "SLIMGPU_Evaluate": (C.c_int32, [C.c_int32, C.c_int32, i32_1d, i32_1d, C.c_void_p, i32_1d,
                                     C.c_int32, f64_1d, i32_1d]),
    "SLIMGPU_LastStats": (C.c_int32, [C.POINTER(Stats)]),
    "SLIMGPU_LastColumnStats": (C.c_int32, [C.c_int32] + [C.c_void_p] * 6),
    "SLIMGPU_DeviceCount": (C.c_int32, []),
    "SLIMGPU_LastError": (C.c_char_p, []),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """Return the ctypes handle of libslim.so with every prototype attached."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "slim_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc, gfx950). There is no CPU fallback for SLIM training." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        if os.environ.get("SLIM_AMD_LIB") and not hasattr(lib, name):
            continue  # an older build used for an A/B run may lack the newest entry points
        fn = getattr(lib, name)  # AttributeError here = ABI drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().SLIMGPU_LastError()
    return msg.decode("utf-8", "replace") if msg else ""
