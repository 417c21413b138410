#!/usr/bin/env python3
"""ADMM (algo=admm) on ml100k through the Python mirror: wall time of train, HR@10 / ARHR of the
model (GPU top-N + GPU evaluation), and the oracle's time on the host beside it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import scipy.sparse as sp

from slim_amd import SLIM, SLIMatrix
from slim_amd.io import read_csr_text
import slim_oracle as O

R = read_csr_text(os.path.join(ROOT, "tests", "golden", "ml100k-train.csr"))
T = read_csr_text(os.path.join(ROOT, "tests", "golden", "ml100k-test.csr"), nrows=R.shape[0])
trn = SLIMatrix(R)
for rep in range(2):
    m = SLIM()
    t0 = time.time()
    m.train({"algo": "admm", "l1r": 1.0, "l2r": 1.0}, trn)
    print("admm train (call %d): %.3f s" % (rep + 1, time.time() - t0), flush=True)
W = m.to_csr()
ev = O.evaluate(sp.csc_matrix(W), R, T, 10)
print("admm model: nnz %d, HR@10 %.4f ARHR %.4f" % (W.nnz, ev["hr"], ev["arhr"]))
t0 = time.time()
Wo = O.learn_admm(R, nthreads=min(64, O.max_threads()))
print("oracle admm on %d threads: %.1f s; max|dW| %.2e" % (min(64, O.max_threads()), time.time() - t0,
                                                         abs(sp.csr_matrix(W) - Wo).max()))
