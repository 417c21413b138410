#!/usr/bin/env python3
"""cd_gramr_kernel<10,3> (50 000 - 106 496 items) on a matrix that solves in seconds: the packed item-space
kernel against the tile kernel and the float item-space kernel.  NC=<items> in the environment;
arguments: a sequence of d1 (packed) / f (float G).  (Round 5: the register-load form and the
two-ahead ring of this instantiation failed this check and were removed.)"""
import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slim_amd.engine import DeviceMatrix, KERNEL_GRAM, KERNEL_TILE
def _random_ratings(nu, ni, density, seed):
    rng = np.random.default_rng(seed)
    R = sp.random(nu, ni, density=density, format="csr", random_state=rng, dtype=np.float32)
    R.data = rng.integers(1, 6, R.nnz).astype(np.float32)
    R.sort_indices()
    return R
def maxdiff(a, b):
    d = abs(sp.csc_matrix(a) - sp.csc_matrix(b))
    return float(d.max()) if d.nnz else 0.0
NC = int(os.environ.get('NC', '60000'))
R = _random_ratings(12000, NC, 0.002 * 60000 / NC, 5); R.data[:] = 1.0
m = DeviceMatrix.from_scipy(R, binary=True)
Wt, st = m.learn(seed=2, kernel=KERNEL_TILE, cluster=1); ct = m.column_stats()
print("tile nnz", Wt.nnz, "sweeps", st["sweeps"])
modes = {"d1": dict(), "f": dict(SLIM_GPU_NO_GRAMR="1")}
for name in sys.argv[1:]:
    env = modes[name]
    os.environ.update(env)
    W, s = m.learn(seed=2, kernel=KERNEL_GRAM); c = m.column_stats()
    print(name, "nnz", W.nnz, "sweeps", s["sweeps"], "rows", s["gram_rows"], "maxdiff vs tile", maxdiff(W, Wt), "sweeps same", (c.sweeps == ct.sweeps).mean(), flush=True)
    for k in env: del os.environ[k]
