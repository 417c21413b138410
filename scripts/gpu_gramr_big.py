#!/usr/bin/env python3
"""Item-space CD at full size on the GPU box: the packed-G / on-chip-g kernel (cd_gramr.hpp) against
the float kernels (cd_gram.hpp) on the same columns -- models, sweeps, timings, byte model.
  c4 [ncols]  : the benchmark's first step (default 8192 columns), both kernels
  c4all       : all 100 000 columns, new kernel only
  c5 [npairs] : the first pairs of the C5 grid (warm start), both kernels
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stage(workload, seed=1, ratings=False):
    """ratings: values 1-5 (SURVEY 8(d)'s second C4 run) instead of the binary matrix"""
    import torch
    from slim_amd import synth
    from slim_amd.engine import DeviceMatrix
    dev = torch.device("cuda", 0)
    nrows, ncols, target = synth.CONFIGS[workload]
    rowptr, rowind, rowval = synth.generate_csr(nrows, ncols, target, seed=seed, ratings=ratings, device=dev)
    torch.cuda.synchronize()
    return DeviceMatrix.from_device_ptrs(nrows, ncols, rowptr.data_ptr(), rowind.data_ptr(),
                                         rowval.data_ptr() if ratings else 0,
                                         keepalive=(rowptr, rowind, rowval), device=0)


def maxdiff(a, b):
    d = abs(sp.csc_matrix(a) - sp.csc_matrix(b))
    return float(d.max()) if d.nnz else 0.0


def run(mat, tag, **kw):
    t0 = time.perf_counter()
    W, st = mat.learn(**kw)
    dt = time.perf_counter() - t0
    cs = mat.column_stats()
    print("%s: wall %.2f s, kernel %.2f s, G %.2f s, rows %d, bytes %.3e -> %.0f GB/s by its model, nnzW %d, sweeps %d"
          % (tag, dt, st["kernel_ms"] * 1e-3, st["gram_build_ms"] * 1e-3, st["gram_rows"], st["gram_bytes"],
             st["gram_bytes"] / max(st["kernel_ms"], 1e-9) / 1e6, st["nnzW"], st["sweeps"]), flush=True)
    return W, st, cs


def main():
    from slim_amd.engine import KERNEL_GRAM
    what = sys.argv[1] if len(sys.argv) > 1 else "c4"
    kw = dict(l1r=1.0, l2r=1.0, optTol=1e-7, niters=10000, seed=1, kernel=KERNEL_GRAM)
    if what == "c4":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
        ratings = "--ratings" in sys.argv
        mat = stage("c4", ratings=ratings)
        Wn, sn, cn = run(mat, "c4%s %d columns, packed G / g on chip" % (" (ratings 1-5)" if ratings else "", n),
                         col_begin=0, col_end=n, **kw)
        Wn2, sn2, _ = run(mat, "   again (G there)", col_begin=0, col_end=n, **kw)
        print("   same model twice: %s" % (maxdiff(Wn, Wn2) == 0.0))
        if "--no-float" not in sys.argv:
            os.environ["SLIM_GPU_NO_GRAMR"] = "1"
            Wf, sf, cf = run(mat, "   float kernel (cd_gram_kernel<8,0>)", col_begin=0, col_end=n, **kw)
            del os.environ["SLIM_GPU_NO_GRAMR"]
            print("   packed vs float: max|dW| %.3e, sweeps same %.4f, D %d/%d U %d/%d" % (
                maxdiff(Wn, Wf), (cn.sweeps == cf.sweeps).mean(), cn.D.sum(), cf.D.sum(), cn.U.sum(), cf.U.sum()))
    elif what == "c4all":
        ratings = "--ratings" in sys.argv
        mat = stage("c4", ratings=ratings)
        run(mat, "c4%s all columns, packed G / g on chip" % (" (ratings 1-5)" if ratings else ""), **kw)
    elif what == "c5":
        npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
        mat = stage("c5")
        pairs = [tuple(map(float, ln.split())) for ln in open(os.path.join(ROOT, "tests", "golden", "l12file")) if ln.strip()]
        for env in ("", "1"):
            if env:
                os.environ["SLIM_GPU_NO_GRAMR"] = "1"
            prev = None
            models = []
            for l1, l2 in pairs[:npairs]:
                W, st, cs = run(mat, "c5 pair (%g, %g) %s" % (l1, l2, "float kernel" if env else "packed"),
                                imodel=prev, **dict(kw, l1r=l1, l2r=l2))
                prev = W
                models.append((W, cs.sweeps.copy()))
            if env:
                del os.environ["SLIM_GPU_NO_GRAMR"]
                for k, ((Wa, sa), (Wb, sb)) in enumerate(zip(first, models)):
                    print("   pair %d packed vs float: max|dW| %.3e sweeps same %.4f" % (k, maxdiff(Wa, Wb), (sa == sb).mean()))
            first = models


if __name__ == "__main__":
    main()
