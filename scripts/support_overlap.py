#!/usr/bin/env python3
"""How much do the 32 problems of a tile share?  For tiles of the cost-ordered work list of a
synthetic configuration: |support| per problem, the union over the tile, and the sharing factor
sum / union -- the number of problems a row of G fetched once could serve (cd_gram.hpp reads the
row once per problem).  Also: active-set sizes, sweeps, and the share of a tile's updates that a
batch of B consecutive visits would see (the visiting order is a random permutation of the union
of the active sets, so a batch holds B / nunion of every support)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--begin", type=int, default=8192 * 3)
    ap.add_argument("--columns", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    from slim_amd import synth
    from slim_amd.engine import KERNEL_GRAM, DeviceMatrix
    dev = torch.device("cuda", 0)
    nrows, ncols, target = synth.CONFIGS[args.workload]
    rowptr, rowind, _ = synth.generate_csr(nrows, ncols, target, seed=args.seed, device=dev)
    mat = DeviceMatrix.from_device_ptrs(nrows, ncols, rowptr.data_ptr(), rowind.data_ptr(), 0,
                                        keepalive=(rowptr, rowind), device=0)
    b = args.begin
    W, st = mat.learn(col_begin=b, col_end=b + args.columns, kernel=KERNEL_GRAM, l1r=1.0, l2r=1.0,
                      optTol=1e-7, niters=10000, seed=args.seed)
    cs = mat.column_stats()
    cost = mat.column_cost()
    cols = np.arange(b, b + args.columns)
    order = cols[np.argsort(-cost[cols], kind="stable")]
    W = W.tocsc()
    colnnz = np.diff(mat.column_view()[0])
    print("kernel %.1f ms, G %.1f ms, rows %d, nnzW %d" % (st["kernel_ms"], st["gram_build_ms"], st["gram_rows"], W.nnz))
    pop_rank = np.empty(ncols, np.int64)
    pop_rank[np.argsort(-colnnz, kind="stable")] = np.arange(ncols)
    tot_sum = tot_union = 0
    for t in range(0, len(order), 32):
        items = order[t:t + 32]
        sup = [W.indices[W.indptr[i]:W.indptr[i + 1]] for i in items]
        s = sum(len(x) for x in sup)
        u = np.unique(np.concatenate(sup)) if s else np.zeros(0, np.int64)
        # how many problems share a support row
        cnt = np.bincount(np.concatenate(sup), minlength=ncols) if s else np.zeros(ncols, np.int64)
        tot_sum += s
        tot_union += len(u)
        if (t // 32) % max(1, (len(order) // 32) // 8) == 0:
            print("tile %4d: col nnz %7d..%7d  na %6d..%6d  sweeps %2d..%2d  |supp| %5d..%5d  sum %7d  union %6d  "
                  "share %.2f  rows in >=16 problems %5d  median pop-rank of the union %6d" % (
                      t // 32, colnnz[items].min(), colnnz[items].max(), cs.nacols[items].min(), cs.nacols[items].max(),
                      cs.sweeps[items].min(), cs.sweeps[items].max(), min(len(x) for x in sup), max(len(x) for x in sup),
                      s, len(u), s / max(1, len(u)), int((cnt >= 16).sum()),
                      int(np.median(pop_rank[u])) if len(u) else -1), flush=True)
    print("all %d tiles: sum %d union %d sharing factor %.2f" % (len(order) // 32, tot_sum, tot_union, tot_sum / max(1, tot_union)))


if __name__ == "__main__":
    main()
