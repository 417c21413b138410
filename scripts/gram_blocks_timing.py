#!/usr/bin/env python3
"""G = R^T R in row blocks at full size (one device): seconds to form the block a rank of N would
form (SLIMGPU_MatrixGramBuildRows), for N = 8, 4, 2, 1, and to commit (byte planes).
usage: gram_blocks_timing.py [c4|c5]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    import torch
    from gpu_gramr_big import stage
    from slim_amd.distributed import gram_blocks
    what = sys.argv[1] if len(sys.argv) > 1 else "c4"
    mat = stage(what)
    for world in (8, 4, 2, 1):
        for rank in sorted({0, world - 1}):
            b, e = gram_blocks(mat.ncols, world)[rank]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mat.gram_build_rows(b, e)
            torch.cuda.synchronize()
            print("%s: block %d of %d (rows %d..%d): %.2f s" % (what, rank, world, b, e, time.perf_counter() - t0), flush=True)
    t0 = time.perf_counter()
    mat.gram_commit()
    torch.cuda.synchronize()
    print("%s: commit (byte planes): %.2f s" % (what, time.perf_counter() - t0))


if __name__ == "__main__":
    main()
