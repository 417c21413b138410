#!/usr/bin/env python3
"""Model-selection grid on a synthetic configuration with R resident in HBM (SURVEY.md §8(d) C5;
reference loop: src/programs/slim_mselect.c:94-113).  Solves the first N (l1, l2) pairs of
tests/golden/l12file in file order, each warm-started from the previous model, and prints
item-columns/s per pair.

  python scripts/c5_grid.py [--workload c5] [--pairs 4] [--scale 1.0] [--columns 0]
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
    ap.add_argument("--workload", default="c5")
    ap.add_argument("--pairs", type=int, default=4)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--columns", type=int, default=0, help="solve only the first N columns (0 = all)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cluster", type=int, default=0, help="tile cluster size (0 = automatic)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "tile", "gram"],
                    help="auto: the engine's choice (item-space CD once the grid is announced)")
    ap.add_argument("--no-announce", action="store_true",
                    help="do not tell the engine how many solves are coming (SLIMGPU_MatrixExpectSolves)")
    ap.add_argument("--resident", default="", choices=["", "fetch", "nofetch"],
                    help="models stay in HBM (SLIMGPU_LearnResident), each pair warm-started from the resident "
                         "previous one; fetch: every model still reaches the host, copied beside the next solve")
    args = ap.parse_args()
    import torch
    from slim_amd import synth
    from slim_amd import _lib as _l
    from slim_amd.engine import KERNEL_AUTO, KERNEL_GRAM, KERNEL_TILE, DeviceMatrix

    dev = torch.device("cuda", 0)
    nrows, ncols, target = synth.scaled(args.workload, args.scale) if args.scale != 1 \
        else synth.CONFIGS[args.workload]
    rowptr, rowind, _ = synth.generate_csr(nrows, ncols, target, seed=args.seed, device=dev)
    torch.cuda.synchronize()
    mat = DeviceMatrix.from_device_ptrs(nrows, ncols, rowptr.data_ptr(), rowind.data_ptr(), 0,
                                        keepalive=(rowptr, rowind), device=0)
    pairs = [tuple(map(float, l.split())) for l in open(os.path.join(ROOT, "tests", "golden", "l12file"))
             if l.strip()][:args.pairs]
    ce = args.columns or mat.ncols
    kernel = {"auto": KERNEL_AUTO, "tile": KERNEL_TILE, "gram": KERNEL_GRAM}[args.kernel]
    if not args.no_announce:
        mat.expect_solves(len(pairs))
    prev = None
    out = []
    t_all = time.time()
    if args.resident:
        fetch = args.resident == "fetch"
        fetched_nnz = []
        for l1, l2 in pairs:
            t0 = time.time()
            cur, st = mat.learn_resident(warm=prev, l1r=l1, l2r=l2, optTol=1e-7, niters=10000, seed=args.seed,
                                         col_begin=0, col_end=ce, kernel=kernel)
            t1 = time.time()
            if prev is not None:
                if fetch:   # begun before this solve: the copy ran beside it
                    h = prev.fetch(return_handle=True)
                    v = C.cast(h, C.POINTER(_l.CsrView)).contents
                    fetched_nnz.append(int(v.colptr[v.ncols]))
                    mat._lib.SLIM_FreeModel(C.byref(C.c_void_p(h)))
                prev.free()
            if fetch:
                cur.fetch_begin()
            prev = cur
            dt = time.time() - t0
            rec = {"l1": l1, "l2": l2, "columns": ce, "wall_s": round(dt, 2), "solve_call_s": round(t1 - t0, 2),
                   "kernel_s": round(st["kernel_ms"] * 1e-3, 2), "columns_per_s": round(ce / dt, 1),
                   "nnzW": int(st["nnzW"]), "kernel": int(st["kernel"]),
                   "gram_build_s": round(st["gram_build_ms"] * 1e-3, 2), "gather_s": round(st["gather_ms"] * 1e-3, 2),
                   "warm": len(out) > 0, "resident": args.resident}
            out.append(rec)
            print(json.dumps(rec), flush=True)
        t0 = time.time()
        if fetch:
            h = prev.fetch(return_handle=True)
            v = C.cast(h, C.POINTER(_l.CsrView)).contents
            fetched_nnz.append(int(v.colptr[v.ncols]))
            mat._lib.SLIM_FreeModel(C.byref(C.c_void_p(h)))
            assert fetched_nnz == [r["nnzW"] for r in out], "a fetched model does not have the solve's nnz"
        prev.free()
        print(json.dumps({"workload": args.workload, "nrows": nrows, "ncols": ncols, "nnz": int(rowind.numel()),
                          "pairs": len(out), "resident": args.resident, "last_fetch_s": round(time.time() - t0, 2),
                          "total_s": round(time.time() - t_all, 1)}))
        return
    for l1, l2 in pairs:
        t0 = time.time()
        h, st = mat.learn(imodel=prev, return_handle=True, l1r=l1, l2r=l2, optTol=1e-7, niters=10000,
                          seed=args.seed, col_begin=0, col_end=ce, kernel=kernel,
                          **({"cluster": args.cluster} if args.cluster else {}))
        dt = time.time() - t0
        if prev is not None:
            mat._lib.SLIM_FreeModel(C.byref(C.c_void_p(prev)))
        prev = h
        cs = mat.column_stats()
        rec = {"l1": l1, "l2": l2, "columns": ce, "wall_s": round(dt, 2),
               "kernel_s": round(st["kernel_ms"] * 1e-3, 2), "columns_per_s": round(ce / dt, 1),
               "alg_GBps": round(st["alg_bytes"] / (st["kernel_ms"] * 1e-3) / 1e9, 1),
               "nnzW": int(st["nnzW"]), "mean_sweeps": round(float(cs.sweeps[:ce].mean()), 2),
               "kernel": int(st["kernel"]), "gram_build_s": round(st["gram_build_ms"] * 1e-3, 2),
               "gather_s": round(st["gather_ms"] * 1e-3, 2), "updates_per_col": round(float(cs.U[:ce].sum()) / max(float(cs.D[:ce].sum()), 1.0), 4),
               "warm": len(out) > 0}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    print(json.dumps({"workload": args.workload, "nrows": nrows, "ncols": ncols, "nnz": int(rowind.numel()),
                      "pairs": len(out), "total_s": round(time.time() - t_all, 1)}))


if __name__ == "__main__":
    main()
