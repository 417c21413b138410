#!/usr/bin/env python3
"""does forming tiles by co-rating similarity shrink the union of the 32 active
sets a residual-kernel tile walks?  CPU only.  For ml100k and Automotive (data with neighbourhood
structure; tests/golden) and a 1/10-scale C4 at 0.1 % density (the generator: none), all tiles of
(a) the engine's cost-ordered work list and (b) greedy similarity tiles (an unassigned seed item
plus the 31 unassigned items most co-rated with it): nnz-weighted |union| / mean |own active set|
= bytes a tile gathers over bytes its problems need."""
import os
import sys
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slim_amd.io import read_csr_text  # noqa: E402


def ratio(R, tiles, l1):
    Rc = sp.csc_matrix(R)
    nnz_col = np.diff(Rc.indptr).astype(np.int64)
    A = (R.T @ R).toarray()            # aTy of every item (small matrices only)
    np.fill_diagonal(A, 0)
    act = A > l1
    tot_u = tot_o = 0.0
    for t in tiles:
        sets = act[:, t]
        tot_u += nnz_col[sets.any(axis=1)].sum()
        tot_o += np.mean([nnz_col[sets[:, k]].sum() for k in range(len(t))])
    return tot_u / max(tot_o, 1.0)


def tilings(R):
    Rc = sp.csc_matrix(R)
    deg = np.diff(R.indptr).astype(np.int64)
    ncols = R.shape[1]
    cost = np.zeros(ncols, np.int64)
    np.add.at(cost, np.repeat(np.arange(ncols), np.diff(Rc.indptr)), deg[Rc.indices])
    order = np.argsort(-cost, kind="stable")
    by_cost = [order[g:g + 32] for g in range(0, ncols, 32)]
    C = (sp.csr_matrix(R.T) @ R).toarray().astype(np.float64)   # co-rating counts
    np.fill_diagonal(C, -1)
    free = np.ones(ncols, bool)
    by_sim = []
    for seed in order:
        if not free[seed]:
            continue
        free[seed] = False
        c = np.where(free, C[seed], -2.0)
        near = np.argsort(-c)[:31]
        near = near[c[near] > -2.0]
        free[near] = False
        by_sim.append(np.concatenate([[seed], near]))
    return by_cost, by_sim


def main():
    import torch  # noqa: F401
    from slim_amd import synth
    data = []
    data.append(("ml100k", read_csr_text(os.path.join(ROOT, "tests", "golden", "ml100k-train.csr")), 1.0))
    try:
        from slim_amd.io import read_ijv
        t = read_ijv(os.path.join(ROOT, "tests", "golden", "AutomotiveTrain.ijv"))
        data.append(("Automotive", sp.csr_matrix((t[:, 2].astype(np.float32),
                                                  (t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)))), 1.0))
    except Exception as e:   # noqa: BLE001
        print("Automotive skipped:", e)
    nr, nc, nz = synth.scaled("c4-0.1pct", 0.1)
    rp, ri, _ = synth.generate_csr(nr, nc, nz, seed=1, device="cpu")
    data.append(("c4 at 0.1 %%, 1/10 scale (%d x %d)" % (nr, nc),
                 sp.csr_matrix((np.ones(ri.numel(), np.float32), ri.numpy(), rp.numpy()), shape=(nr, nc)), 1.0))
    for name, R, l1 in data:
        R = sp.csr_matrix(R)
        R.data[:] = 1.0 if name.startswith("c4") else R.data
        by_cost, by_sim = tilings(R)
        print("%s (%d x %d, %d nnz): nnz-weighted union / own, cost-ordered tiles %.3f, similarity tiles %.3f"
              % (name, R.shape[0], R.shape[1], R.nnz, ratio(R, by_cost, l1), ratio(R, by_sim, l1)), flush=True)


if __name__ == "__main__":
    main()
