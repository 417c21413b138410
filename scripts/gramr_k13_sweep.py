#!/usr/bin/env python3
"""Differential sweep of cd_gramr_kernel<10,3> (57 345 - 106 496 items): the packed item-space
kernel against the float item-space kernel (same fmaf sequence: max|dW| must be 0, same sweeps for
every column), optionally against the tile kernel too.

  gramr_k13_sweep.py [--items 82000,94000,106000] [--density 0.5,1,2] [--seeds 1,2,3]
                     [--users 4000] [--tile] [--detail] [--repeat 2] [--warm]

density is in units of 1e-3.  --detail: for a failing case, which columns differ and at which
popularity ranks (group = rank >> 13: groups 0-9 live in registers, 10-12 in LDS).  --repeat: the
packed solve is run that many times (a race shows as run-to-run differences).  SLIM_AMD_LIB selects
an A/B build (scripts/build_variant.sh)."""
import argparse, os, sys, time, json
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slim_amd.engine import DeviceMatrix, KERNEL_GRAM, KERNEL_TILE


def ratings(nu, ni, density, seed, skew):
    rng = np.random.default_rng(seed)
    if skew <= 0:
        R = sp.random(nu, ni, density=density, format="csr", random_state=rng, dtype=np.float32)
    else:  # popularity ~ 1 / (rank + c)^skew over a random permutation of the ids
        nnz = int(nu * ni * density)
        p = 1.0 / (np.arange(ni) + 20.0) ** skew
        p /= p.sum()
        cols = rng.permutation(ni)[rng.choice(ni, size=nnz, p=p)]
        rows = rng.integers(0, nu, nnz)
        R = sp.csr_matrix((np.ones(nnz, np.float32), (rows, cols)), shape=(nu, ni))
        R.sum_duplicates()
    R.data[:] = 1.0
    R.sort_indices()
    return R


def coldiff(a, b):
    d = abs(sp.csc_matrix(a) - sp.csc_matrix(b)).tocsc()
    per = np.zeros(a.shape[1])
    if d.nnz:
        per = np.maximum.reduceat(np.r_[d.data, 0.0], np.minimum(d.indptr[:-1], d.nnz)) * (np.diff(d.indptr) > 0)
    return d, per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", default="82000,94000,106000")
    ap.add_argument("--density", default="0.5,1,2")
    ap.add_argument("--seeds", default="1,2,3")
    ap.add_argument("--users", type=int, default=4000)
    ap.add_argument("--skew", type=float, default=0.0)
    ap.add_argument("--tile", action="store_true")
    ap.add_argument("--detail", action="store_true")
    ap.add_argument("--warm", action="store_true")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--l1", type=float, default=1.0)
    ap.add_argument("--l2", type=float, default=1.0)
    a = ap.parse_args()
    print("lib", os.environ.get("SLIM_AMD_LIB", "slim_amd/libslim.so"), flush=True)
    bad = 0
    for ni in [int(x) for x in a.items.split(",")]:
        for dn in [float(x) for x in a.density.split(",")]:
            for seed in [int(x) for x in a.seeds.split(",")]:
                R = ratings(a.users, ni, dn * 1e-3, seed, a.skew)
                m = DeviceMatrix.from_scipy(R, binary=True)
                kw = dict(seed=seed + 1, kernel=KERNEL_GRAM, l1r=a.l1, l2r=a.l2)
                os.environ["SLIM_GPU_NO_GRAMR"] = "1"
                t0 = time.time()
                Wf, sf = m.learn(**kw)
                cf = m.column_stats()
                tf = time.time() - t0
                del os.environ["SLIM_GPU_NO_GRAMR"]
                rec = dict(items=ni, users=a.users, density=dn * 1e-3, seed=seed, nnz=int(R.nnz), W_nnz=int(Wf.nnz),
                           float_s=round(tf, 2))
                Ws = []
                for rep in range(a.repeat):
                    t0 = time.time()
                    W, s = m.learn(**kw)
                    c = m.column_stats()
                    rec["packed_s"] = round(time.time() - t0, 2)
                    d, per = coldiff(W, Wf)
                    rec["max_dW_vs_float" + ("" if rep == 0 else "_run%d" % rep)] = float(per.max()) if per.size else 0.0
                    rec["bad_columns" + ("" if rep == 0 else "_run%d" % rep)] = int((per > 0).sum())
                    rec["sweeps_same"] = bool((c.sweeps == cf.sweeps).all())
                    Ws.append(W)
                if a.repeat > 1:
                    rec["runs_identical"] = all(abs(Ws[0] - w).nnz == 0 for w in Ws[1:])
                if a.warm:  # a second solve warm-started from the first, both forms
                    os.environ["SLIM_GPU_NO_GRAMR"] = "1"
                    Wf2, _ = m.learn(imodel=Wf, **dict(kw, l2r=a.l2 * 2))
                    del os.environ["SLIM_GPU_NO_GRAMR"]
                    W2, _ = m.learn(imodel=Wf, **dict(kw, l2r=a.l2 * 2))
                    _, per2 = coldiff(W2, Wf2)
                    rec["warm_max_dW_vs_float"] = float(per2.max()) if per2.size else 0.0
                if a.tile:
                    Wt, st = m.learn(seed=seed + 1, kernel=KERNEL_TILE, cluster=1, l1r=a.l1, l2r=a.l2)
                    _, pert = coldiff(Ws[0], Wt)
                    rec["max_dW_vs_tile"] = float(pert.max())
                ok = rec["max_dW_vs_float"] == 0.0 and rec["sweeps_same"] and rec.get("warm_max_dW_vs_float", 0.0) == 0.0
                rec["ok"] = ok
                bad += 0 if ok else 1
                print(json.dumps(rec), flush=True)
                if a.detail and not ok:
                    d, per = coldiff(Ws[0], Wf)
                    nnzc = np.diff(R.tocsc().indptr)
                    order = np.lexsort((np.arange(ni), -nnzc))  # rank -> item (most ratings first, ties by id)
                    rank = np.empty(ni, np.int64)
                    rank[order] = np.arange(ni)
                    bc = np.nonzero(per > 0)[0]
                    print("  bad columns: %d of %d; first ids %s" % (bc.size, ni, bc[:12].tolist()))
                    print("  ranks of the bad columns (the problem's own item): groups", np.bincount(rank[bc] >> 13, minlength=13).tolist())
                    rows = d.indices  # item ids of the differing coefficients
                    print("  ranks of the differing coefficients: groups", np.bincount(rank[rows] >> 13, minlength=13).tolist(),
                          " thread (rank>>4)&511 / 64 (wave):", np.bincount(((rank[rows] >> 4) & 511) >> 6, minlength=8).tolist())
                    print("  |dW| quantiles", np.quantile(d.data, [0.5, 0.9, 0.99, 1.0]).tolist())
                    for j in bc[:3]:
                        col = d[:, j]
                        print("  column %d (rank %d): %d differing, sweeps packed %d float %d; ranks %s" %
                              (j, rank[j], col.nnz, c.sweeps[j], cf.sweeps[j], np.sort(rank[col.indices])[:16].tolist()))
                m.close()
    print("FAILED CASES:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
