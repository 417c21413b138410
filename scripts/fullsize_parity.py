#!/usr/bin/env python3
"""Full-size parity probe (GPU box): whole 32-column tiles of the C4 / C5 synthetic matrices
solved by the tile kernel and by the oracle walking the same tile order, plus the order-noise
envelope (oracle vs oracle in two visiting orders) on the same columns.

  python scripts/fullsize_parity.py --workload c4 --tiles median,sampled,heavy

Prints one JSON object per tile; tests/test_fullsize_parity.py asserts the same comparisons.
The oracle is the checker here, never the thing measured.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def maxdiff(a, b):
    d = abs(a - b)
    return float(d.max()) if d.nnz else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--tiles", default="median,sampled")
    ap.add_argument("--batch-begin", type=int, default=0)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--noise", action="store_true", help="also oracle-vs-oracle order noise")
    ap.add_argument("--tight", action="store_true", help="also optTol 1e-12 vs ORDER_PERM")
    args = ap.parse_args()

    import numpy as np
    import scipy.sparse as sp
    import torch
    import slim_oracle as O
    from slim_amd import synth
    from slim_amd.engine import DeviceMatrix

    dev = torch.device("cuda", 0)
    nrows, ncols, target = synth.CONFIGS[args.workload]
    t0 = time.time()
    rowptr, rowind, _ = synth.generate_csr(nrows, ncols, target, seed=args.seed, device=dev)
    torch.cuda.synchronize()
    print("generated %dx%d nnz %d in %.1f s" % (nrows, ncols, rowind.numel(), time.time() - t0),
          flush=True)
    mat = DeviceMatrix.from_device_ptrs(nrows, ncols, rowptr.data_ptr(), rowind.data_ptr(), 0,
                                        keepalive=(rowptr, rowind), device=0)
    cost = mat.column_cost()
    t0 = time.time()
    R = sp.csr_matrix((np.ones(rowind.numel(), np.float32), rowind.cpu().numpy(),
                       rowptr.cpu().numpy()), shape=(nrows, ncols))
    print("host copy %.1f s" % (time.time() - t0), flush=True)
    threads = args.threads or min(32, O.max_threads())

    b = args.batch_begin
    cols = np.arange(b, b + args.batch)
    order = cols[np.argsort(-cost[cols], kind="stable")]  # the engine's work list of the batch
    ntiles = len(order) // 32
    picks = {}
    for name in args.tiles.split(","):
        if name == "median":
            picks[name] = ntiles // 2
        elif name == "heavy":
            picks[name] = 0
        elif name == "light":
            picks[name] = ntiles - 1
        elif name == "sampled":
            # the tile holding the first column bench.py's cpu_baseline samples from this batch
            rng = np.random.default_rng(args.seed)
            c0 = int(np.sort(b + rng.permutation(args.batch)[:8])[0])
            picks[name] = int(np.where(order == c0)[0][0]) // 32
        else:
            picks[name] = int(name)
    kw = dict(l1r=1.0, l2r=1.0, optTol=1e-7)
    for name, g in picks.items():
        tile = np.ascontiguousarray(order[g * 32:(g + 1) * 32], dtype=np.int32)
        out = {"workload": args.workload, "tile": name, "tile_index": g,
               "columns": [int(tile.min()), int(tile.max())], "threads": threads}
        t0 = time.time()
        W, st = mat.learn(columns=tile, niters=10000, seed=args.seed, **kw)
        out["gpu_s"] = round(time.time() - t0, 2)
        cs = mat.column_stats()
        out["gpu_sweeps_max"] = int(cs.sweeps[tile].max())
        out["W_nnz"] = int(W.nnz)
        out["W_max"] = float(W.data.max()) if W.nnz else 0.0
        t0 = time.time()
        Wo, so, _, _ = O.learn_cd_tile(R, tileP=32, order=tile, maxniters=10000, seed=args.seed,
                                       nthreads=threads, binary=True, return_stats=True, **kw)
        out["oracle_tile_s"] = round(time.time() - t0, 1)
        out["max_abs_dW_tile_order"] = maxdiff(W[:, tile], Wo[:, tile])
        out["same_sweeps"] = float((cs.sweeps[tile] == so["sweeps"][tile]).mean())
        out["same_nacols"] = bool((cs.nacols[tile] == so["nacols"][tile]).all())
        out["same_G"] = bool((cs.G[tile] == so["G"][tile]).all())
        out["same_D"] = float((cs.D[tile] == so["D"][tile]).mean())
        out["W_nnz_oracle"] = int(Wo[:, tile].nnz)
        print(json.dumps(out), flush=True)
        if args.noise:
            t0 = time.time()
            Wp = O.learn_cd(R, cols=tile, order=O.ORDER_PERM, seed=args.seed, aty=O.ATY_GRAM,
                            maxniters=10000, nthreads=threads, binary=True, chunk=1, **kw)
            out["oracle_perm_s"] = round(time.time() - t0, 1)
            out["noise_oracle_tile_vs_perm"] = maxdiff(Wo[:, tile], Wp[:, tile])
            out["max_abs_dW_gpu_vs_perm"] = maxdiff(W[:, tile], Wp[:, tile])
            Wl = O.learn_cd(R, cols=tile, order=O.ORDER_LOCAL, seed=args.seed + 7, aty=O.ATY_GRAM,
                            maxniters=10000, nthreads=threads, binary=True, chunk=1, **kw)
            out["noise_oracle_perm_vs_local"] = maxdiff(Wl[:, tile], Wp[:, tile])
            print(json.dumps(out), flush=True)
        if args.tight:
            kt = dict(l1r=1.0, l2r=1.0, optTol=1e-12)
            Wt, _ = mat.learn(columns=tile, niters=100000, seed=args.seed, **kt)
            t0 = time.time()
            Wpt = O.learn_cd(R, cols=tile, order=O.ORDER_PERM, seed=args.seed, aty=O.ATY_GRAM,
                             maxniters=100000, nthreads=threads, binary=True, chunk=1, **kt)
            out["tight_oracle_s"] = round(time.time() - t0, 1)
            out["tight_gpu_sweeps_max"] = int(mat.column_stats().sweeps[tile].max())
            out["tight_max_abs_dW_gpu_vs_perm"] = maxdiff(Wt[:, tile], Wpt[:, tile])
            print(json.dumps(out), flush=True)
    mat.close()


if __name__ == "__main__":
    main()
