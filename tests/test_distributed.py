"""The N > 1 path on CPU: world_size-2 gloo processes run the sharding driver of
slim_amd/distributed.py (broadcast of R, cost-balanced column blocks, gather of the learned
columns).  The per-rank solve is played by the oracle here (no GPU on this box); on GPUs the
same driver calls DeviceMatrix.learn."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN, ROOT
from slim_amd.distributed import partition_columns


def test_partition_columns_balances_cost():
    rng = np.random.default_rng(0)
    cost = rng.pareto(1.2, 5000) * 100
    for world in (1, 2, 4, 8):
        blocks = partition_columns(cost, world)
        assert blocks[0][0] == 0 and blocks[-1][1] == 5000 and len(blocks) == world
        assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        loads = np.array([(cost[b:e] + 1).sum() for b, e in blocks])
        assert loads.max() <= loads.mean() + (cost.max() + 1)   # within one column of ideal
    assert partition_columns(np.zeros(3), 8)[-1][1] == 3        # more ranks than columns
    assert partition_columns([], 2) == [(0, 0), (0, 0)]


def test_gram_blocks_cover_the_items_once():
    """Row blocks of G = R^T R for N ranks (build_gram_sharded): contiguous, in order, equal to one row,
    empty blocks only behind the last item."""
    from slim_amd.distributed import gram_blocks
    for ncols, world in ((100000, 8), (20000, 8), (1683, 4), (5, 8), (0, 2), (7, 1)):
        blocks = gram_blocks(ncols, world)
        assert len(blocks) == world and blocks[0][0] == 0 and blocks[-1][1] == ncols
        assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        sizes = [e - b for b, e in blocks]
        assert all(s >= 0 for s in sizes) and sum(sizes) == ncols
        nonempty = [s for s in sizes if s > 0]
        assert sizes == sorted(sizes, reverse=True)      # full blocks first, then a shorter one, then empty ones


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OracleMatrix(object):
    """Stands in for slim_amd.engine.DeviceMatrix on a GPU-less box."""

    def __init__(self, R):
        import slim_oracle as O
        self.O, self.R = O, R
        self.ncols = R.shape[1]

    def column_cost(self):
        Rc = self.R.tocsc()
        deg = np.diff(self.R.indptr)
        return np.array([deg[Rc.indices[Rc.indptr[c]:Rc.indptr[c + 1]]].sum()
                         for c in range(self.ncols)], dtype=np.int64)

    def learn(self, col_begin=0, col_end=None, seed=1, shard=None, **kw):
        cols = np.arange(col_begin, self.ncols if col_end is None else col_end, dtype=np.int32)
        if shard is not None:   # the engine's shards: granules of 32 of the cost-ordered list
            index, count = shard
            order = self.O.tile_work_order(self.R, col_begin, self.ncols if col_end is None else col_end)
            cols = np.sort(order[(np.arange(order.size) // 32) % count == index]).astype(np.int32)
        W, st, err, obj = self.O.learn_cd(self.R, order=self.O.ORDER_PERM, seed=seed,
                                          aty=self.O.ATY_GRAM, cols=cols, return_stats=True)
        return W, {"ncols_solved": len(cols), "objval": obj, "error": err, "nnzW": W.nnz}


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    from slim_amd.distributed import broadcast_csr, gather_model, learn_sharded
    from slim_amd.io import read_csr_text
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if rank == 0:
            R = read_csr_text(os.path.join(GOLDEN, "ml100k-train.csr"))[:, :400].tocsr()
            R.sort_indices()
            ptr = torch.from_numpy(R.indptr.astype(np.int64))
            ind = torch.from_numpy(R.indices.astype(np.int32))
            val = torch.from_numpy(R.data.astype(np.float32))
        else:
            ptr = ind = val = None
        ptr, ind, val = broadcast_csr(ptr, ind, val, src=0)
        R = sp.csr_matrix((val.numpy(), ind.numpy(), ptr.numpy()))
        Ws, stats_s, (si, sc) = learn_sharded(_OracleMatrix(R), seed=3)      # shards (default)
        assert (si, sc) == (rank, world)
        W, stats, (b, e) = learn_sharded(_OracleMatrix(R), seed=3, partition="blocks")
        assert abs(sp.csc_matrix(Ws) - sp.csc_matrix(W)).nnz == 0
        assert stats_s["totals"]["ncols_solved"] == stats["totals"]["ncols_solved"]
        sp.save_npz(os.path.join(out_dir, "w%d.npz" % rank), sp.csc_matrix(W))
        np.save(os.path.join(out_dir, "b%d.npy" % rank), np.array([b, e, stats["ncols_solved"]]))
        np.save(os.path.join(out_dir, "t%d.npy" % rank),
                np.array([stats["totals"][k] for k in ("objval", "error", "nnzW", "ncols_solved")]))
        # binary matrices broadcast without a value array
        p2, i2, v2 = broadcast_csr(ptr if rank == 0 else None, ind if rank == 0 else None, None)
        assert v2 is None and torch.equal(p2, ptr) and torch.equal(i2, ind)
        # a rank with an empty block still takes part in the gather
        n = W.shape[0]
        part = sp.csc_matrix(W)[:, :n] if rank == 0 else sp.csc_matrix((n, n), dtype=np.float32)
        full = gather_model(part)
        assert abs(full - W).max() == 0
        only0 = gather_model(part, dst=0)
        assert (only0 is None) == (rank != 0)
        if rank == 0:
            assert abs(only0 - W).max() == 0
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_learn_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    import slim_oracle as O
    from slim_amd.io import read_csr_text
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    R = read_csr_text(os.path.join(GOLDEN, "ml100k-train.csr"))[:, :400].tocsr()
    want, _, err, obj = O.learn_cd(R, order=O.ORDER_PERM, seed=3, aty=O.ATY_GRAM, return_stats=True)
    w0 = sp.load_npz(str(tmp_path / "w0.npz"))
    w1 = sp.load_npz(str(tmp_path / "w1.npz"))
    assert abs(w0 - w1).nnz == 0                  # every rank holds the full model
    assert abs(w0 - want).max() == 0              # and it is the single-process result
    b0, b1 = np.load(str(tmp_path / "b0.npy")), np.load(str(tmp_path / "b1.npy"))
    assert b0[0] == 0 and b0[1] == b1[0] and b1[1] == want.shape[1]
    assert b0[2] + b1[2] == want.shape[1] and min(b0[2], b1[2]) > 0
    # the objective / error reductions (estimate.c:371-373) are summed over the ranks
    t0, t1 = np.load(str(tmp_path / "t0.npy")), np.load(str(tmp_path / "t1.npy"))
    assert np.array_equal(t0, t1)
    assert abs(t0[0] - obj) <= 1e-9 * obj and abs(t0[1] - err) <= 1e-9 * err
    assert t0[2] == want.nnz and t0[3] == want.shape[1]
