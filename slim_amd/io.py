"""Readers/writers for the on-disk formats that sit either side of the training path.

The parsing itself lives in GKlib in the reference (gk_csr_Read/gk_csr_Write,
call sites /root/reference/src/programs/slim_learn.c:27,83 and
src/libslim/pyapi.c:49,61); layouts are taken from the shipped sample files and
those call sites.
"""
import numpy as np
import scipy.sparse as sp


def read_csr_text(path, readvals=True, nrows=None):
    """'csr' text format (test/ml100k-*.csr): one line per user, whitespace
    separated ``item value`` pairs (``readvals=False``: bare item ids, the
    'csrnv' format).  Ids are used as written (gk_csr_Read(..., numbering=0),
    slim_learn.c:27), so 1-based files produce an empty column 0 and
    ncols = max id + 1.  Blank lines are empty rows."""
    indptr = [0]
    indices = []
    data = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if readvals:
                if len(tok) % 2:
                    raise ValueError("odd number of tokens on a csr line in %s" % path)
                indices.extend(int(t) for t in tok[0::2])
                data.extend(float(t) for t in tok[1::2])
            else:
                indices.extend(int(t) for t in tok)
            indptr.append(len(indices))
    if nrows is not None:
        while len(indptr) < nrows + 1:
            indptr.append(indptr[-1])
    indices = np.asarray(indices, dtype=np.int32)
    vals = (np.asarray(data, dtype=np.float32) if readvals
            else np.ones(len(indices), dtype=np.float32))
    ncols = int(indices.max()) + 1 if indices.size else 0
    return sp.csr_matrix((vals, indices, np.asarray(indptr, dtype=np.int64)),
                         shape=(len(indptr) - 1, ncols))


def write_csr_text(path, mat, writevals=True):
    """Inverse of :func:`read_csr_text` (Py_csr_save / slim_learn model output:
    row view, ``item weight`` pairs).  Values are written with repr-exact
    precision (%.9g) so a save/load round trip is lossless for float32."""
    mat = sp.csr_matrix(mat)
    with open(path, "w") as f:
        for r in range(mat.shape[0]):
            lo, hi = mat.indptr[r], mat.indptr[r + 1]
            if writevals:
                f.write(" ".join("%d %.9g" % (c, v) for c, v in
                                 zip(mat.indices[lo:hi], mat.data[lo:hi])))
            else:
                f.write(" ".join("%d" % c for c in mat.indices[lo:hi]))
            f.write("\n")


def read_ijv(path, delimiter=None):
    """'ijv' triplets (test/Automotive*.ijv): ``user item value`` per line.
    Returns a float64 (n,3) array -- what ``pandas.read_csv(...).values`` gives
    the reference's Python wrapper (python-package/test/main.py:31-32)."""
    return np.loadtxt(path, delimiter=delimiter, dtype=np.float64, ndmin=2)


def read_l12file(path):
    """l12file: one ``l1 l2`` pair per line (slim_mselect.c:100-101)."""
    pairs = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if len(tok) >= 2:
                pairs.append((float(tok[0]), float(tok[1])))
    return pairs
