#!/usr/bin/env python3
"""Warm-started grid step A/B on one box (slim_mselect.c:99-113): pair 1 of
tests/golden/l12file cold, then pair 2 from that model once per variant of the environment
(fold:xcd:gram = SLIM_GPU_FOLD row | col, SLIM_GPU_XCD 1 | 0, screen-sum cache 1 | 0; a fourth
field, the per-XCD fold token of profiles/r03/warm_step_ab.txt, existed while that experiment did) -- the variants differ in nothing but how the
previous coefficients are folded into the residual and where the cluster members sit.

  python scripts/warm_ab.py [--workload c5] [--columns 0] [--variants row:1,row:0,col:1]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c5")
    ap.add_argument("--columns", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--variants", default="row:1:1,row:0:1,col:1:1,row:1:0")
    ap.add_argument("--cold-tol", type=float, default=1e-7)
    args = ap.parse_args()
    import torch
    from slim_amd import synth
    from slim_amd.engine import DeviceMatrix

    dev = torch.device("cuda", 0)
    nrows, ncols, target = synth.CONFIGS[args.workload]
    rowptr, rowind, _ = synth.generate_csr(nrows, ncols, target, seed=args.seed, device=dev)
    torch.cuda.synchronize()
    mat = DeviceMatrix.from_device_ptrs(nrows, ncols, rowptr.data_ptr(), rowind.data_ptr(), 0,
                                        keepalive=(rowptr, rowind), device=0)
    pairs = [tuple(map(float, l.split())) for l in open(os.path.join(ROOT, "tests", "golden", "l12file"))
             if l.strip()]
    ce = args.columns or mat.ncols

    def solve(l1, l2, prev, tag):
        t0 = time.time()
        h, st = mat.learn(imodel=prev, return_handle=True, l1r=l1, l2r=l2, optTol=args.cold_tol,
                          niters=10000, seed=args.seed, col_begin=0, col_end=ce)
        dt = time.time() - t0
        cs = mat.column_stats()
        rec = {"tag": tag, "l1": l1, "l2": l2, "columns": ce, "wall_s": round(dt, 2),
               "kernel_s": round(st["kernel_ms"] * 1e-3, 2), "columns_per_s": round(ce / dt, 1),
               "alg_GBps": round(st["alg_bytes"] / (st["kernel_ms"] * 1e-3) / 1e9, 1),
               "nnzW": int(st["nnzW"]), "mean_sweeps": round(float(cs.sweeps[:ce].mean()), 3),
               "objval": st["objval"]}
        print(json.dumps(rec), flush=True)
        return h

    first = solve(pairs[0][0], pairs[0][1], None, "cold")
    for v in args.variants.split(","):
        f = v.split(":")
        fold, xcd = f[0], f[1]
        gram = f[-1] if len(f) > 2 else "1"
        os.environ["SLIM_GPU_FOLD"] = fold
        os.environ["SLIM_GPU_XCD"] = xcd
        if gram == "0":
            os.environ["SLIM_GPU_NO_GRAM"] = "1"
        else:
            os.environ.pop("SLIM_GPU_NO_GRAM", None)
        h = solve(pairs[1][0], pairs[1][1], first,
                  "warm fold=%s xcd=%s gram=%s" % (fold, xcd, gram))
        import ctypes as C
        mat._lib.SLIM_FreeModel(C.byref(C.c_void_p(h)))
    # an l1 step from the same model (several sweeps), default settings
    for k in ("SLIM_GPU_FOLD", "SLIM_GPU_XCD", "SLIM_GPU_NO_GRAM"):
        os.environ.pop(k, None)
    solve(pairs[9][0], pairs[9][1], first, "warm l1 step (defaults)")


if __name__ == "__main__":
    main()
