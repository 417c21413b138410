"""Full-size parity (BASELINE.json configs[3] and [4]): whole 32-column tiles of the 1M x 100K
(~1e9 nnz) and 10M x 20K (~1e9 nnz) synthetic matrices, generated on the device from the
benchmark's seed, solved by the tile kernel through the C ABI (SLIMGPU_LearnColumns) and by the
oracle walking the same tile in the same visiting order (oracle_learn_cd_tile: the reference's
fp64 three-pass arithmetic, cd.c:101-142 / estimate.c:402-530, with the engine's ShuffleList).

What VERDICT r1 asked to explain -- bench.py's cpu_baseline.max_abs_dW_vs_gpu = 8.2e-3 on the
driver's run -- was a bookkeeping error of bench.py (it compared the columns sampled from the
FIRST timed step with the model returned by the LAST step, which holds other columns, so the
figure was max|W| itself); the kernels agree with the oracle to ~5e-9 at full size
(profiles/r02/fullsize_parity.txt), and the order-to-order envelope on these columns is 1.4e-5.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import slim_oracle as O
from slim_amd.engine import KERNEL_GRAM, KERNEL_TILE, DeviceMatrix

pytestmark = pytest.mark.gpu


def maxdiff(a, b):
    d = abs(a - b)
    return float(d.max()) if d.nnz else 0.0


def _stage(workload, seed=1):
    import torch
    from slim_amd import synth
    dev = torch.device("cuda", 0)
    nrows, ncols, target = synth.CONFIGS[workload]
    rowptr, rowind, _ = synth.generate_csr(nrows, ncols, target, seed=seed, device=dev)
    mat = DeviceMatrix.from_device_ptrs(nrows, ncols, rowptr.data_ptr(), rowind.data_ptr(), 0,
                                        keepalive=(rowptr, rowind), device=0)
    R = sp.csr_matrix((np.ones(rowind.numel(), np.float32), rowind.cpu().numpy(),
                       rowptr.cpu().numpy()), shape=(nrows, ncols))
    assert abs(R.nnz - target) <= 0.01 * target      # nnz ~ 1e9 after de-duplication
    return mat, R


def _batch_tiles(mat, begin, batch):
    """The engine's tiles of a bench step: the batch's columns by descending cost, 32 at a time."""
    cost = mat.column_cost()
    cols = np.arange(begin, begin + batch)
    return cols[np.argsort(-cost[cols], kind="stable")].astype(np.int32).reshape(-1, 32)


def _check_tile(mat, R, tile, threads, geoms=({},)):
    """One whole tile through the C ABI, once per entry of `geoms` (kernel / cluster options), each
    against ONE oracle walk of the same tile in the same visiting order."""
    kw = dict(l1r=1.0, l2r=1.0, optTol=1e-7)
    Wo, so, _, _ = O.learn_cd_tile(R, tileP=32, order=tile, maxniters=10000, seed=1,
                                   nthreads=threads, binary=True, return_stats=True, **kw)
    out = []
    for geom in geoms:
        W, st = mat.learn(columns=tile, niters=10000, seed=1, **kw, **geom)
        cs = mat.column_stats()
        assert W[:, tile].nnz > 0 and W.nnz == W[:, tile].nnz
        assert np.array_equal(cs.nacols[tile], so["nacols"][tile])      # identical active sets
        assert np.array_equal(cs.G[tile], so["G"][tile])
        assert (cs.sweeps[tile] == so["sweeps"][tile]).mean() >= 0.98   # identical sweep counts
        assert maxdiff(W[:, tile], Wo[:, tile]) <= 2e-5                 # stated fp32 tolerance
        out.append((W, st))
    return out, Wo


def _check_tile_tight(mat, R, tile, threads, geoms=({},)):
    """at optTol 1e-12 the visiting order no longer matters -- the kernels
    against the oracle in its own per-item order (reference arithmetic), <= 2e-5."""
    kt = dict(l1r=1.0, l2r=1.0, optTol=1e-12)
    Wp = O.learn_cd(R, cols=tile, order=O.ORDER_PERM, seed=1, aty=O.ATY_GRAM, maxniters=100000,
                    nthreads=threads, binary=True, chunk=1, **kt)
    for geom in geoms:
        Wt, _ = mat.learn(columns=tile, niters=100000, seed=1, **kt, **geom)
        assert maxdiff(Wt[:, tile], Wp[:, tile]) <= 2e-5            # observed 2.6e-8 / 1.0e-7


@pytest.mark.timeout(1500, method="thread")
def test_c4_full_size_tiles_match_oracle_in_tile_order():
    """C4, seed 1.  (1) The product default: SLIMGPU_Learn with default options over the
    benchmark's first step (8192 columns) on a fresh handle -- the engine's own choice must be
    item space (G = R^T R of all 100 000 items built inside the call, cd_gram*.hpp), and its
    median tile is compared with the oracle walking that tile of that work list.  (2) Whole
    tiles through SLIMGPU_LearnColumns on BOTH paths (residual kernel, item-space kernel) against
    one oracle walk each: the median tile of the step and the tile holding the first column
    bench.py's cpu_baseline samples; the optTol 1e-12 check on both paths; and -- so that the
    heavy phase and clusters of 16 are in play -- the same tiles once more as a two-tile launch
    of the residual kernel with a heavy phase."""
    mat, R = _stage("c4")
    O.cache_setup(True)           # one transpose of R for the oracle calls of this test
    threads = min(32, O.max_threads())
    tiles = _batch_tiles(mat, 0, 8192)   # bench.py DEFAULT_BATCH
    kw = dict(l1r=1.0, l2r=1.0, optTol=1e-7)
    # (0) a first solve of ONE tile: G would never pay for itself -- the engine stays in user space
    _, st0 = mat.learn(columns=tiles[len(tiles) // 2], niters=10000, seed=1, **kw)
    assert st0["kernel"] == KERNEL_TILE
    # (1) default options, the benchmark's step
    Wd, std = mat.learn(col_begin=0, col_end=8192, niters=10000, seed=1, **kw)
    csd = mat.column_stats()
    assert std["kernel"] == KERNEL_GRAM and std["gram_build_ms"] > 0
    k = len(tiles) // 2
    Wk, sk, _, _ = O.learn_cd_tile(R, tileP=32, order=tiles.reshape(-1), tiles=(k, 1), maxniters=10000,
                                   seed=1, nthreads=threads, binary=True, return_stats=True, **kw)
    tk = tiles[k]
    assert np.array_equal(csd.nacols[tk], sk["nacols"][tk])
    assert (csd.sweeps[tk] == sk["sweeps"][tk]).mean() >= 0.98
    assert maxdiff(Wd[:, tk], Wk[:, tk]) <= 2e-5
    # (2) whole tiles on both paths
    rng = np.random.default_rng(1)
    c0 = int(np.sort(rng.permutation(8192)[:8])[0])
    sampled = int(np.where(tiles == c0)[0][0])
    picked = [tiles[len(tiles) // 2], tiles[sampled]]
    both_paths = (dict(kernel=KERNEL_TILE), dict(kernel=KERNEL_GRAM))
    for t in picked:                                    # one tile: clusters of 16 (residual kernel)
        (pair, _) = _check_tile(mat, R, t, threads, geoms=both_paths)
        assert pair[0][1]["kernel"] == KERNEL_TILE and pair[1][1]["kernel"] == KERNEL_GRAM
        assert maxdiff(pair[0][0][:, t], pair[1][0][:, t]) <= 2e-5
    _check_tile_tight(mat, R, picked[0], threads, geoms=both_paths)
    # both tiles in one launch, the first one as a "heavy" tile on a cluster of 16, the second
    # on clusters of 4: the per-problem arithmetic and the visiting order may not depend on it
    both = np.concatenate(picked)
    cost = mat.column_cost()
    order = both[np.argsort(-cost[both], kind="stable")]
    W2, _ = mat.learn(columns=both, niters=10000, seed=1, kernel=KERNEL_TILE,
                      cluster=4, heavy_tiles=1, heavy_cluster=16, **kw)
    Wo2 = O.learn_cd_tile(R, tileP=32, order=order, maxniters=10000, seed=1, nthreads=threads,
                          binary=True, **kw)
    assert maxdiff(W2[:, both], Wo2[:, both]) <= 2e-5
    Wg2, _ = mat.learn(columns=both, niters=10000, seed=1, kernel=KERNEL_GRAM, **kw)
    assert maxdiff(Wg2[:, both], Wo2[:, both]) <= 2e-5
    O.cache_setup(False)
    mat.close()


@pytest.mark.timeout(900, method="thread")
def test_c5_full_size_tile_matches_oracle_in_tile_order():
    """C5 (10M x 20K, tall-skinny), seed 1: the median tile of a 4096-column step."""
    mat, R = _stage("c5")
    threads = min(32, O.max_threads())
    tiles = _batch_tiles(mat, 0, 4096)
    O.cache_setup(True)
    _check_tile(mat, R, tiles[len(tiles) // 2], threads, geoms=(dict(kernel=KERNEL_TILE),))
    _check_tile_tight(mat, R, tiles[len(tiles) // 2], threads, geoms=(dict(kernel=KERNEL_TILE),))
    O.cache_setup(False)
    mat.close()


@pytest.mark.timeout(1200, method="thread")
def test_c5_full_size_warm_started_grid_steps_match_oracle():
    """What config 5 is about (VERDICT r2 missing #2): consecutive (l1, l2) pairs of
    test/l12file, each solved from the previous model (slim_mselect.c:99-113,
    estimate.c:453-471, cd.c:108-110), on one whole tile of the 10M x 20K matrix -- slices six
    chunks long, clusters of 16.  Pair 1 cold, pair 2 (an l2 step: one sweep) and an l1 step
    (the active sets shrink, several sweeps) warm-started, GPU and oracle each from their own
    previous model, in the tile's visiting order: <= 2e-5, identical active sets and sweep
    counts.  Both forms of the fold (row-wise: the default; column-wise) are checked -- and the
    same three steps in item space (KERNEL_GRAM: G = R^T R of the whole 10M x 20K matrix built
    once, the tile's 32 problems solved on it, each step from that path's own previous model),
    which is the path a 45-pair grid takes."""
    import os
    mat, R = _stage("c5")
    threads = min(32, O.max_threads())
    tiles = _batch_tiles(mat, 0, 4096)
    tile = tiles[len(tiles) // 2]
    pairs = [tuple(map(float, ln.split())) for ln in
             open(os.path.join(os.path.dirname(__file__), "golden", "l12file")) if ln.strip()]
    assert pairs[0] == (0.1, 0.1) and pairs[1] == (0.1, 0.5) and pairs[9] == (0.5, 0.1)
    O.cache_setup(True)
    prev_g = prev_o = prev_i = None
    for step, (l1, l2) in enumerate((pairs[0], pairs[1], pairs[9])):
        kw = dict(l1r=l1, l2r=l2, optTol=1e-7)
        W, st = mat.learn(columns=tile, niters=10000, seed=1, imodel=prev_g, kernel=KERNEL_TILE, **kw)
        cs = mat.column_stats()
        sweeps_g, na_g = cs.sweeps[tile].copy(), cs.nacols[tile].copy()
        Wo, so, _, _ = O.learn_cd_tile(R, tileP=32, order=tile, maxniters=10000, seed=1,
                                       nthreads=threads, binary=True, return_stats=True,
                                       imodel=prev_o, **kw)
        assert W[:, tile].nnz > 0
        assert np.array_equal(na_g, so["nacols"][tile])
        assert (sweeps_g == so["sweeps"][tile]).mean() >= 0.98
        assert maxdiff(W[:, tile], Wo[:, tile]) <= 2e-5
        # the same step in item space
        Wi, si = mat.learn(columns=tile, niters=10000, seed=1, imodel=prev_i, kernel=KERNEL_GRAM, **kw)
        ci = mat.column_stats()
        assert si["kernel"] == KERNEL_GRAM and (si["gram_build_ms"] > 0) == (step == 0)
        assert np.array_equal(ci.nacols[tile], so["nacols"][tile])
        assert (ci.sweeps[tile] == so["sweeps"][tile]).mean() >= 0.98
        assert maxdiff(Wi[:, tile], Wo[:, tile]) <= 2e-5
        assert abs(si["objval"] - st["objval"]) <= 1e-4 * st["objval"]
        prev_i = Wi
        if step == 1:
            assert sweeps_g.max() <= 2          # an l2 step of 0.4 moves nothing: one sweep
            assert ci.sweeps[tile].max() <= 2
            os.environ["SLIM_GPU_FOLD"] = "col"  # the other fold: the same step again
            try:
                Wc, _ = mat.learn(columns=tile, niters=10000, seed=1, imodel=prev_g, kernel=KERNEL_TILE, **kw)
            finally:
                del os.environ["SLIM_GPU_FOLD"]
            assert np.array_equal(mat.column_stats().sweeps[tile], sweeps_g)
            assert maxdiff(Wc[:, tile], Wo[:, tile]) <= 2e-5
        prev_g, prev_o = W, Wo
    O.cache_setup(False)
    mat.close()


@pytest.mark.timeout(900, method="thread")
def test_c5_full_size_grid_step_from_the_carried_g_matches_oracle():
    """The grid as Py_SLIM_Mselect runs it since round 6: ALL 20 000 columns of the 10M x 20K matrix per
    pair, the models resident in HBM (SLIMGPU_LearnResident), and the second pair -- an l2 step -- started
    from the g the first solve left on chip instead of folding the model again (cd_gramr.hpp: g_load).
    One whole tile of that launch (the median of its work list, keyed by its position) against the
    oracle walking that tile in the same visiting order, cold and warm-started from ITS previous model
    (estimate.c:453-464, cd.c:108-110: the fold the engine skipped): <= 2e-5, identical active sets and
    sweep counts; and the carried solve streams no fold rows."""
    import os
    mat, R = _stage("c5")
    threads = min(32, O.max_threads())
    order = _batch_tiles(mat, 0, mat.ncols)         # the engine's work list of an all-column solve
    k = len(order) // 2
    tile = order[k]
    pairs = [tuple(map(float, ln.split())) for ln in
             open(os.path.join(os.path.dirname(__file__), "golden", "l12file")) if ln.strip()]
    assert pairs[0] == (0.1, 0.1) and pairs[1] == (0.1, 0.5)
    mat.expect_solves(len(pairs))
    O.cache_setup(True)
    prev_d = prev_o = None
    rows = []
    for step, (l1, l2) in enumerate(pairs[:2]):
        kw = dict(l1r=l1, l2r=l2, optTol=1e-7)
        cur, st = mat.learn_resident(warm=prev_d, niters=10000, seed=1, **kw)
        assert st["kernel"] == KERNEL_GRAM
        cs = mat.column_stats()
        W = cur.fetch()
        rows.append(int(st["gram_rows"]))
        Wo, so, _, _ = O.learn_cd_tile(R, tileP=32, order=order.reshape(-1), tiles=(k, 1), maxniters=10000,
                                       seed=1, nthreads=threads, binary=True, return_stats=True,
                                       imodel=prev_o, **kw)
        assert W[:, tile].nnz > 0
        assert np.array_equal(cs.nacols[tile], so["nacols"][tile])
        assert (cs.sweeps[tile] == so["sweeps"][tile]).mean() >= 0.98
        assert maxdiff(W[:, tile], Wo[:, tile]) <= 2e-5
        if prev_d is not None:
            prev_d.free()
        prev_d, prev_o = cur, Wo
        if step == 1:
            assert cs.sweeps[tile].max() <= 2
            # one row of G per coefficient that moved, none for the fold (the folded step reads ~2 per coefficient)
            assert rows[1] <= 1.1 * W.nnz
    O.cache_setup(False)
    mat.close()
