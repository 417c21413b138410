#!/usr/bin/env python3
"""ADMM (algo=admm, estimate.c:38-304) once at its own configuration: BASELINE.json configs[4]'s
matrix (10M users x 20K items, ~1e9 nnz), m = 20 000 -- six dense m x m fp64 matrices = 19 GB.
Prints the stage times the library reports under SLIM_GPU_TRACE=1 (R^T R, Cholesky, inverse,
the dgemm chain of the 30 iterations) and the dgemm rate against the fp64 MFMA peak.

  SLIM_GPU_TRACE=1 python scripts/admm_c5.py [--scale 1.0]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c5")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--l1", type=float, default=1.0)
    ap.add_argument("--l2", type=float, default=1.0)
    args = ap.parse_args()
    import numpy as np
    import torch
    from slim_amd import _lib, synth
    from slim_amd.engine import make_options
    from slim_amd.constants import Opt

    dev = torch.device("cuda", 0)
    nrows, ncols, target = synth.scaled(args.workload, args.scale) if args.scale != 1 \
        else synth.CONFIGS[args.workload]
    rowptr, rowind, _ = synth.generate_csr(nrows, ncols, target, seed=1, device=dev)
    ptr = np.ascontiguousarray(rowptr.cpu().numpy().astype(np.intp))
    ind = np.ascontiguousarray(rowind.cpu().numpy().astype(np.int32))
    del rowptr, rowind
    torch.cuda.empty_cache()
    lib = _lib.load()
    iopt, dopt = make_options(l1r=args.l1, l2r=args.l2)
    iopt[Opt.ALGO] = 0  # SLIM_ALGO_ADMM (slim.h:190-193: admm = 0, cd = 1)
    st = C.c_int32(0)
    t0 = time.time()
    h = lib.SLIM_Learn(nrows, ptr, ind, None, iopt.ctypes.data_as(C.c_void_p),
                       dopt.ctypes.data_as(C.c_void_p), None, C.byref(st))
    dt = time.time() - t0
    if not h:
        raise SystemExit("SLIM_Learn(algo=admm) failed (%d): %s" % (st.value, _lib.last_error()))
    view = C.cast(h, C.POINTER(_lib.CsrView)).contents
    m = view.ncols
    nnz = view.rowptr[view.nrows]
    flops = 30 * 2.0 * m ** 3          # one m^3 dgemm per iteration (estimate.c:169-213: T = P W + A)
    print("admm %dx%d nnz %d: m = %d, SLIM_Learn %.1f s, model nnz %d; the 30 iteration dgemms are "
          "%.2e flop (AMD's published fp64 matrix peak, 78.6 TFLOP/s -- the guide lists none -> %.1f s at peak)"
          % (nrows, ncols, ind.size, m, dt, nnz, flops, flops / 78.6e12), flush=True)
    hh = C.c_void_p(h)
    lib.SLIM_FreeModel(C.byref(hh))


if __name__ == "__main__":
    main()
