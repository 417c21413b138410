#!/usr/bin/env python3
"""Where a problem's cycles go in cd_gramr_kernel (a library built with -DSLIM_GRAMR_PROF=1:
scripts/build_variant.sh prof "-DSLIM_GRAMR_PROF=1", run with SLIM_AMD_LIB=variants/libslim_prof.so).
That build reports, in place of D / U / bytes / rows of a column, the shader cycles wavefront 0 spent
  fetch : top of a batch to the end of fetch_g (two barriers, the export of 64 entries of g)
  decide: picking the next mover (and, at a batch's end, finding none)
  apply : streaming and applying rows;   first: of that, until group 0 of a row had landed
usage: gramr_prof.py [c4|c5] [ncols]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    from gpu_gramr_big import stage
    from slim_amd.engine import KERNEL_GRAM
    what = sys.argv[1] if len(sys.argv) > 1 else "c4"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    if what == "small":  # 60 000 items (the <10,3> kernel) on a matrix that solves in seconds
        import numpy as np
        import scipy.sparse as sp
        from slim_amd.engine import DeviceMatrix
        rng = np.random.default_rng(5)
        R = sp.random(12000, 60000, density=0.002, format="csr", random_state=rng, dtype=np.float32)
        R.data[:] = 1.0
        R.sort_indices()
        mat = DeviceMatrix.from_scipy(R, binary=True)
    else:
        mat = stage(what)
    kw = dict(l1r=1.0, l2r=1.0, optTol=1e-7, niters=10000, seed=1, kernel=KERNEL_GRAM, col_begin=0, col_end=n)
    mat.learn(**kw)
    W, st = mat.learn(**kw)
    cs = mat.column_stats()
    fetch, dec, app = float(cs.D[:n].sum()), float(cs.U[:n].sum()), float(st["gram_bytes"])
    first = 256.0 * float(st["gram_rows"])
    export = 256.0 * float(cs.nacols[:n].astype("float64").sum())
    tot = fetch + dec + app
    print("%s %d columns: kernel %.2f s; cycles of wavefront 0 over all problems: fetch %.3e (%.1f %%), decide %.3e (%.1f %%), "
          "apply %.3e (%.1f %%) of which waiting for a row's first group %.3e (%.1f %% of all); of fetch, before the barrier %.3e (%.1f %% of all); sweeps %d"
          % (what, n, st["kernel_ms"] * 1e-3, fetch, 100 * fetch / tot, dec, 100 * dec / tot, app, 100 * app / tot,
             first, 100 * first / tot, export, 100 * export / tot, st["sweeps"]))
    print("   cycles per problem %.3e -> at 256 problems at a time: %.2f s at 2.4 GHz" % (tot / n, tot / 256 / 2.4e9))


if __name__ == "__main__":
    main()
