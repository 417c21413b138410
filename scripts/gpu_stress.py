"""Determinism stress of the clustered tile kernel: same inputs, same cluster size => bitwise
identical W, counters and losses on every repetition."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp
from slim_amd.engine import DeviceMatrix, KERNEL_TILE, KERNEL_TILE16
from slim_amd.io import read_csr_text

R = read_csr_text(os.path.join(ROOT, "tests/golden/ml100k-train.csr"))
mat = DeviceMatrix.from_scipy(R)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
for kern in (KERNEL_TILE, KERNEL_TILE16):
    for cl in (1, 2, 4, 8):
        ref = None
        for rep in range(reps):
            W, st = mat.learn(seed=1, kernel=kern, cluster=cl)
            cs = mat.column_stats()
            sig = (W.nnz, float(W.data.astype(np.float64).sum()), st["objval"], int(cs.sweeps.sum()), int(cs.nacols.sum()), int(cs.D.sum()))
            if ref is None:
                ref, Wref = sig, W
            elif sig != ref or abs(W - Wref).nnz:
                bad += 1
                d = abs(W - Wref)
                print("MISMATCH kern", kern, "cluster", cl, "rep", rep, sig, "vs", ref, "max|d|", d.max() if d.nnz else 0,
                      "cols differing", np.unique(d.tocoo().col).size if d.nnz else 0)
        print("kern", kern, "cluster", cl, "ok" if bad == 0 else "bad so far %d" % bad, ref)
print("TOTAL MISMATCHES", bad)
