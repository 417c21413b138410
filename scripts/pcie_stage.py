#!/usr/bin/env python3
"""PCIe-inclusive staging of a host CSR (the SLIM_Learn / SLIMGPU_MatrixFromHost boundary):
generate the configuration on the GPU, copy it to (pageable) host memory, then time
SLIMGPU_MatrixFromHost = H2D of rowptr/rowind[/rowval] + device transpose + column norms.

  python scripts/pcie_stage.py [--workload c4] [--ratings]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--ratings", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    from slim_amd import _lib, synth
    from slim_amd.engine import make_options

    dev = torch.device("cuda", 0)
    nrows, ncols, target = synth.CONFIGS[args.workload]
    rowptr, rowind, rowval = synth.generate_csr(nrows, ncols, target, seed=1, ratings=args.ratings,
                                                device=dev)
    ptr = rowptr.cpu().numpy().astype(np.intp)
    ind = rowind.cpu().numpy()
    val = rowval.cpu().numpy() if args.ratings else None
    del rowptr, rowind, rowval
    torch.cuda.empty_cache()
    lib = _lib.load()
    iopt, _ = make_options()
    for rep in range(2):
        st = C.c_int32(0)
        t0 = time.time()
        h = lib.SLIMGPU_MatrixFromHost(nrows, ptr, ind, None if val is None else val.ctypes.data_as(C.c_void_p),
                                       iopt.ctypes.data_as(C.c_void_p), C.byref(st))
        dt = time.time() - t0
        assert h, _lib.last_error()
        nbytes = ptr.nbytes + ind.nbytes + (val.nbytes if val is not None else 0)
        print(json.dumps({"workload": args.workload, "ratings": bool(args.ratings), "nnz": int(ind.size),
                          "host_bytes": nbytes, "stage_s": round(dt, 3),
                          "GBps_incl_transpose": round(nbytes / dt / 1e9, 1), "rep": rep}), flush=True)
        hh = C.c_void_p(h)
        lib.SLIMGPU_MatrixFree(C.byref(hh))


if __name__ == "__main__":
    main()
