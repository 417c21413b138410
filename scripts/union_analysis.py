#!/usr/bin/env python3
"""Why the 0.1 % variant of C4 moves more bytes than it needs (CPU only, ~2 minutes): a tile walks the
UNION of its 32 problems' active sets and gathers a whole 128-byte line per visited nnz, but a
problem only uses the visits of its own active set.  For tiles of the engine's cost-ordered work
list this prints |union| / mean |own set|, plain and weighted by column length (= bytes gathered
over bytes needed), and the same for a tile built from one item and the 31 items most co-rated
with it.  (The matrix is generated on the CPU: not bit-identical to the GPU's seed-1 matrix,
same model.)"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from slim_amd import synth
t0 = time.time()
nrows, ncols, target = synth.CONFIGS["c4-0.1pct"]
rowptr, rowind, _ = synth.generate_csr(nrows, ncols, target, seed=1, device="cpu")
print("generated nnz", rowind.numel(), "in %.0f s" % (time.time() - t0), flush=True)
R = sp.csr_matrix((np.ones(rowind.numel(), np.float32), rowind.numpy(), rowptr.numpy()), shape=(nrows, ncols))
Rc = R.tocsc()
deg = np.diff(R.indptr).astype(np.int64)
colof = np.repeat(np.arange(ncols), np.diff(Rc.indptr))
G = np.zeros(ncols, np.int64); np.add.at(G, colof, deg[Rc.indices])
order = np.argsort(-G, kind="stable")
def active(q):
    a = (R.T @ Rc[:, q]).toarray().ravel()   # co-rating counts with item q
    a[q] = 0
    return a > 1.0                          # estimate.c:433-444 with l1 = 1
rng = np.random.default_rng(0)
for name, tiles in (("cost-ordered tiles (the engine's)", [order[g * 32:(g + 1) * 32] for g in (10, 100, 500, 1500, 2500)]),):
    for t in tiles:
        sets = [active(int(q)) for q in t]
        sizes = np.array([s.sum() for s in sets])
        union = np.logical_or.reduce(sets).sum()
        nnz_col = np.diff(Rc.indptr)
        # nnz-weighted: what the tile gathers (union) against what its problems need (mean of own sets)
        w_union = nnz_col[np.logical_or.reduce(sets)].sum()
        w_own = np.mean([nnz_col[s].sum() for s in sets])
        print("%s: tile of items with %d..%d ratings: active sets %d..%d (mean %.0f), union %d = %.2f x mean; nnz-weighted union / mean own = %.2f"
              % (name, nnz_col[t].min(), nnz_col[t].max(), sizes.min(), sizes.max(), sizes.mean(), union, union / sizes.mean(), w_union / w_own), flush=True)
# the same 32 items against the 31 items most co-rated with the tile's first item
for g in (100, 1500):
    seed = int(order[g * 32])
    c = (R.T @ Rc[:, seed]).toarray().ravel(); c[seed] = -1
    near = np.argsort(-c)[:31]
    t = np.concatenate([[seed], near])
    sets = [active(int(q)) for q in t]
    sizes = np.array([s.sum() for s in sets])
    nnz_col = np.diff(Rc.indptr)
    w_union = nnz_col[np.logical_or.reduce(sets)].sum(); w_own = np.mean([nnz_col[s].sum() for s in sets])
    print("seed %d + its 31 most co-rated items (ratings %d..%d): union %.2f x mean set; nnz-weighted %.2f"
          % (seed, nnz_col[t].min(), nnz_col[t].max(), np.logical_or.reduce(sets).sum() / sizes.mean(), w_union / w_own), flush=True)
print("total %.0f s" % (time.time() - t0))
