#!/usr/bin/env python3
"""Scratch checks of the item-space kernel (cd_gram.hpp) on the GPU box: against the tile kernel
(same visiting order) and the oracle's tile walk, cold and warm, for every workgroup geometry."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import slim_oracle as O  # noqa: E402
from slim_amd.engine import KERNEL_GRAM, KERNEL_TILE, DeviceMatrix  # noqa: E402
from slim_amd.io import read_csr_text  # noqa: E402


def maxdiff(a, b):
    d = abs(sp.csc_matrix(a) - sp.csc_matrix(b))
    return float(d.max()) if d.nnz else 0.0


def rnd(nu, ni, density, seed, binary=False):
    rng = np.random.default_rng(seed)
    R = sp.random(nu, ni, density=density, format="csr", random_state=rng, dtype=np.float32)
    R.data = np.ones(R.nnz, np.float32) if binary else rng.integers(1, 6, R.nnz).astype(np.float32)
    R.sort_indices()
    return R


def check(name, R, oracle=True, binary=False, **kw):
    m = DeviceMatrix.from_scipy(R, binary=binary)
    t0 = time.time()
    Wg, sg = m.learn(kernel=KERNEL_GRAM, **kw)
    tg = time.time() - t0
    csg = m.column_stats()
    Wt, st = m.learn(kernel=KERNEL_TILE, cluster=1, **kw)
    cst = m.column_stats()
    line = "%s: gram kernel %.1f ms (G build %.1f ms, wall %.2f s) tile %.1f ms | gram vs tile %.2e nnz %d/%d " \
        "sweeps same %.4f D %d/%d U %d/%d obj %.6e/%.6e" % (
            name, sg["kernel_ms"], sg["gram_build_ms"], tg, st["kernel_ms"], maxdiff(Wg, Wt), Wg.nnz, Wt.nnz,
            (csg.sweeps == cst.sweeps).mean(), csg.D.sum(), cst.D.sum(), csg.U.sum(), cst.U.sum(),
            sg["objval"], st["objval"])
    assert np.array_equal(csg.nacols, cst.nacols), "active sets differ"
    if oracle:
        Wo, so, err_o, obj_o = O.learn_cd_tile(R, tileP=32, nthreads=8, return_stats=True, binary=binary,
                                               **{k: v for k, v in kw.items() if k in ("seed", "l1r", "l2r")})
        line += " | vs oracle %.2e sweeps %.4f obj %.6e" % (maxdiff(Wg, Wo), (csg.sweeps == so["sweeps"]).mean(), obj_o)
    print(line, flush=True)
    # warm start: from the l1 = 3 model to (1, 0.5)
    first, _ = m.learn(kernel=KERNEL_TILE, cluster=1, **dict(kw, l1r=3.0, l2r=1.0))
    Wg2, sg2 = m.learn(kernel=KERNEL_GRAM, imodel=first, **dict(kw, l1r=1.0, l2r=0.5))
    c2 = m.column_stats()
    Wt2, st2 = m.learn(kernel=KERNEL_TILE, cluster=1, imodel=first, **dict(kw, l1r=1.0, l2r=0.5))
    c3 = m.column_stats()
    print("   warm: gram %.1f ms tile %.1f ms | diff %.2e sweeps same %.4f (%d / %d) obj %.6e/%.6e" % (
        sg2["kernel_ms"], st2["kernel_ms"], maxdiff(Wg2, Wt2), (c2.sweeps == c3.sweeps).mean(),
        c2.sweeps.sum(), c3.sweeps.sum(), sg2["objval"], st2["objval"]), flush=True)
    m.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        check("20000x45000 binary (g in HBM, nw16)", rnd(20000, 45000, 0.002, 8, binary=True), oracle=False, binary=True, seed=2)
        os.environ["SLIM_GPU_GRAM_NW"] = "8"
        check("20000x45000 binary (g in HBM, nw8)", rnd(20000, 45000, 0.002, 8, binary=True), oracle=False, binary=True, seed=2)
        return
    R = read_csr_text(os.path.join(ROOT, "tests", "golden", "ml100k-train.csr"))
    check("ml100k (nw4 v2)", R, seed=1)
    check("60000x96 ratings (nw4)", rnd(60000, 96, 0.08, 11), seed=3)
    check("40000x3000 (nw8 v2)", rnd(40000, 3000, 0.004, 5), oracle=False, seed=2)
    check("40000x6000 binary (nw16 v2)", rnd(40000, 6000, 0.003, 6, binary=True), oracle=False, binary=True, seed=2)
    check("30000x15000 (nw16 v5)", rnd(30000, 15000, 0.002, 7), oracle=False, seed=2)
    check("20000x30000 binary (nw16 v10)", rnd(20000, 30000, 0.002, 8, binary=True), oracle=False, binary=True, seed=2)


if __name__ == "__main__":
    main()
