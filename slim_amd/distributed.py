"""Multi-GPU SLIM training, one process per GPU (the in-process form -- one host thread per
device behind SLIM_Learn, option slot 19 -- lives in csrc/multi_gpu.cpp).

Item columns are independent given the read-only rating matrix
(/root/reference/src/libslim/estimate.c:402-403 is a parallel-for with no
cross-iteration state), so the job shards with NO collective inside the solve:
  1. R is replicated: rank 0's CSR is broadcast once (RCCL over xGMI), every rank
     builds its own column view;
  2. each rank solves its share of the item columns: either shard `rank` of `world` of the
     engine's cost-ordered work list (granules of 32 columns dealt round-robin, so every
     rank gets the same mix of popular and unpopular items and a column's result does not
     depend on the number of ranks), or one contiguous block balanced by the engine's
     per-column cost proxy (partition_columns);
  3. the learned columns are gathered (counts, then padded payload all-gather --
     RCCL has no gatherv) and every rank assembles the full W.
torch.distributed is the transport ("nccl" == RCCL on ROCm, "gloo" for CPU tests).
"""
import numpy as np
import scipy.sparse as sp


def partition_columns(cost, world_size):
    """Contiguous column blocks [(begin, end)] with near-equal summed cost.
    cost: per-column non-negative weights (zero-cost columns are free riders)."""
    cost = np.asarray(cost, dtype=np.float64) + 1.0  # every column costs something
    ncols = cost.size
    if world_size <= 1:
        return [(0, ncols)]
    csum = np.concatenate([[0.0], np.cumsum(cost)])
    targets = csum[-1] * np.arange(1, world_size) / world_size
    cuts = np.searchsorted(csum, targets, side="left")
    cuts = np.clip(cuts, 0, ncols)
    edges = np.concatenate([[0], cuts, [ncols]]).astype(np.int64)
    edges = np.maximum.accumulate(edges)
    return [(int(edges[r]), int(edges[r + 1])) for r in range(world_size)]


def broadcast_csr(rowptr, rowind, rowval, src=0, group=None):
    """Broadcast a CSR held as torch tensors on rank ``src``; other ranks pass None and
    receive freshly allocated tensors on their current device."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    dev = rowptr.device if rank == src else _default_device(group)
    meta = torch.zeros(3, dtype=torch.int64, device=dev)
    if rank == src:
        meta[0], meta[1] = rowptr.numel(), rowind.numel()
        meta[2] = 0 if rowval is None else 1
    dist.broadcast(meta, src, group=group)
    nptr, nnz, has_val = [int(v) for v in meta.tolist()]
    if rank != src:
        rowptr = torch.empty(nptr, dtype=torch.int64, device=dev)
        rowind = torch.empty(nnz, dtype=torch.int32, device=dev)
        rowval = torch.empty(nnz, dtype=torch.float32, device=dev) if has_val else None
    # a few large broadcasts (one per array): at 1e9 nnz, 4 + 4 + 0.008 GB
    dist.broadcast(rowptr, src, group=group)
    dist.broadcast(rowind, src, group=group)
    if has_val:
        dist.broadcast(rowval, src, group=group)
    return rowptr, rowind, rowval


def _default_device(group=None):
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_model(W_local, group=None, dst=None):
    """All-gather column-disjoint pieces of W (scipy CSC, full n x n shape, only this
    rank's columns populated -- a contiguous block or an interleaved shard) into the
    complete model -- on every rank, or (dst given) assembled on rank ``dst`` only; the
    other ranks take part in the collectives and return None."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    W_local = sp.csc_matrix(W_local)
    n = W_local.shape[1]
    if world == 1:
        return W_local
    dev = _default_device(group)
    mine = torch.from_numpy(np.diff(W_local.indptr).astype(np.int32)).to(dev)
    per_rank = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(per_rank, mine, group=group)  # column counts of every rank
    sizes = [int(c.sum()) for c in per_rank]
    pad = max(max(sizes), 1)
    ind = torch.zeros(pad, dtype=torch.int32, device=dev)
    val = torch.zeros(pad, dtype=torch.float32, device=dev)
    ind[:W_local.nnz] = torch.from_numpy(W_local.indices.astype(np.int32)).to(dev)
    val[:W_local.nnz] = torch.from_numpy(W_local.data.astype(np.float32)).to(dev)
    all_ind = [torch.empty_like(ind) for _ in range(world)]
    all_val = [torch.empty_like(val) for _ in range(world)]
    dist.all_gather(all_ind, ind, group=group)
    dist.all_gather(all_val, val, group=group)
    if dst is not None and dist.get_rank(group) != dst:
        return None
    # every column is populated by at most one rank: its entries go to the column's slot of
    # the global CSC, in the order the owner holds them
    counts = [c.cpu().numpy().astype(np.int64) for c in per_rank]
    total = np.sum(counts, axis=0)
    indptr = np.concatenate([[0], np.cumsum(total)])
    indices = np.empty(int(indptr[-1]), np.int32)
    data = np.empty(int(indptr[-1]), np.float32)
    for r in range(world):
        if sizes[r] == 0:
            continue
        local_ptr = np.concatenate([[0], np.cumsum(counts[r])])
        dest = np.repeat(indptr[:-1] - local_ptr[:-1], counts[r]) + np.arange(sizes[r])
        indices[dest] = all_ind[r][:sizes[r]].cpu().numpy()
        data[dest] = all_val[r][:sizes[r]].cpu().numpy()
    return sp.csc_matrix((data, indices, indptr), shape=(n, n))


def gram_blocks(ncols, world_size):
    """Contiguous, equal (to one row) blocks of item ids, one per rank."""
    per = (ncols + world_size - 1) // world_size
    return [(min(r * per, ncols), min((r + 1) * per, ncols)) for r in range(world_size)]


def build_gram_sharded(mat, group=None):
    """G = R^T R of a replicated DeviceMatrix, formed ONCE by the ranks together instead of once
    per rank: rank r forms the rows of block r (SLIMGPU_MatrixGramBuildRows), every block is
    broadcast from its owner straight into the other ranks' buffers (one RCCL broadcast per block:
    blocks may differ by a row, and 5 GB per link is ~50 ms on xGMI), then every rank commits
    (byte planes are formed locally).  Returns the seconds spent (build, exchange, commit).
    With one rank this is the plain build."""
    import time
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    blocks = gram_blocks(mat.ncols, world)
    t0 = time.perf_counter()
    b, e = blocks[rank]
    # A rank whose own block fails (out of memory, say) must not leave the others waiting in a
    # broadcast: the ranks agree on the outcome BEFORE the first collective on G and raise together.
    err = None
    try:
        mat.gram_build_rows(b, e)
        torch.cuda.synchronize()
    except Exception as ex:  # noqa: BLE001 (reported to every rank below)
        err = ex
    t1 = time.perf_counter()
    if world > 1:
        on_device = dist.get_backend(group) == "nccl"   # (gloo: ranks sharing a device in the tests -- host bounce)
        bad = torch.tensor([0 if err is None else 1], dtype=torch.int32,
                           device=torch.device("cuda", mat.device) if on_device else "cpu")
        dist.all_reduce(bad, op=dist.ReduceOp.SUM, group=group)
        if int(bad.item()):
            raise RuntimeError("build_gram_sharded: %d rank(s) could not form their block of G%s"
                               % (int(bad.item()), "" if err is None else " (this rank: %s)" % err))
    elif err is not None:
        raise err
    if world > 1:
        for r, (rb, re) in enumerate(blocks):
            if re <= rb:
                continue
            src = dist.get_global_rank(group, r) if group else r
            block = mat.gram_rows_tensor(rb, re)
            if on_device:
                dist.broadcast(block, src=src, group=group)
            else:
                host = block.cpu() if r == rank else torch.empty(block.shape, dtype=block.dtype)
                dist.broadcast(host, src=src, group=group)
                if r != rank:
                    block.copy_(host)
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    mat.gram_commit()
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2


STAT_KEYS = ("objval", "error", "nnzW", "G", "D", "U", "sweeps", "ncols_solved")


def learn_sharded(mat, group=None, partition="shards", **opts):
    """SLIM_Learn over all ranks of ``group`` on a replicated DeviceMatrix.
    partition: "shards" (rank r solves shard r of the cost-ordered work list; the result of a
    column does not depend on the world size) or "blocks" (contiguous cost-balanced blocks).
    Returns (full W on every rank, this rank's stats + "totals", this rank's (begin, end) for
    blocks or (rank, world) for shards)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if partition == "blocks":
        blocks = partition_columns(mat.column_cost(), world)
        b, e = blocks[rank]
        W_local, stats = mat.learn(col_begin=b, col_end=e, **opts)
    else:
        b, e = rank, world
        W_local, stats = mat.learn(shard=(rank, world), **opts)
    W = gather_model(W_local, group) if world > 1 else W_local
    # the reductions of EstimateModelCD (estimate.c:371-373: error, objval) and the solve
    # counters, summed over the ranks; per-rank values stay under their own names
    stats = dict(stats)
    keys = list(STAT_KEYS)  # the same fixed list on every rank: the all-reduce sizes must match
    if world > 1 and keys:
        import torch
        t = torch.tensor([float(stats.get(k, 0.0)) for k in keys], dtype=torch.float64,
                         device=_default_device(group))
        dist.all_reduce(t, group=group)
        totals = t.tolist()
    else:
        totals = [float(stats.get(k, 0.0)) for k in keys]
    stats["totals"] = dict(zip(keys, totals))
    return W, stats, (b, e)
