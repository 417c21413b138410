#!/usr/bin/env python3
"""GPU probe: slices of one to five chunks per visit (60 000 x 96 valued matrix, ~4800 nnz per
column) through every cluster / heavy-phase geometry, with and without the heavy phase's id
prefetch, against the oracle walking the same tiles.  Found the round-3 bug in which the values
of a block were replaced before the block had been summed (valued matrices, >= 3 chunks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp
import slim_oracle as O
from slim_amd.engine import DeviceMatrix, KERNEL_TILE

def md(a, b):
    d = abs(a - b); return float(d.max()) if d.nnz else 0.0

rng = np.random.default_rng(11)
R = sp.random(60000, 96, density=0.08, format="csr", random_state=rng, dtype=np.float32)
R.data = np.floor(1 + 5 * rng.random(R.nnz)).astype(np.float32)
m = DeviceMatrix.from_scipy(R)
Wo, so, _, _ = O.learn_cd_tile(R, tileP=32, seed=3, nthreads=8, return_stats=True)
for pf in ("1", "0"):
    os.environ["SLIM_GPU_HI_PREFETCH"] = pf
    for geom in (dict(cluster=1, heavy_tiles=0), dict(cluster=2, heavy_tiles=0), dict(cluster=4, heavy_tiles=0),
                 dict(cluster=8, heavy_tiles=0),
                 dict(cluster=1, heavy_tiles=2, heavy_cluster=2), dict(cluster=2, heavy_tiles=1, heavy_cluster=4),
                 dict(cluster=1, heavy_tiles=3, heavy_cluster=4), dict(cluster=1, heavy_tiles=3, heavy_cluster=8)):
        W, st = m.learn(seed=3, kernel=KERNEL_TILE, **geom)
        cs = m.column_stats()
        per_tile = [md(W[:, 32*g:32*g+32][:, :], Wo[:, 32*g:32*g+32]) for g in range(3)]
        print("pf", pf, geom, "maxdiff %.3e" % md(W, Wo), "nnz", W.nnz, Wo.nnz,
              "same sweeps %.2f" % (cs.sweeps == so["sweeps"]).mean(), flush=True)
