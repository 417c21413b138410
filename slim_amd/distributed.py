"""Multi-GPU SLIM training: one process per GPU, item columns block-partitioned.

Item columns are independent given the read-only rating matrix
(/root/reference/src/libslim/estimate.c:402-403 is a parallel-for with no
cross-iteration state), so the job shards with NO collective inside the solve:
  1. R is replicated: rank 0's CSR is broadcast once (RCCL over xGMI), every rank
     builds its own column view;
  2. each rank solves one contiguous block of item columns, blocks balanced by the
     engine's per-column cost proxy;
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
    dev = rowptr.device if rank == src else _default_device()
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


def _default_device():
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_model(W_local, group=None, dst=None):
    """All-gather column-disjoint pieces of W (scipy CSC, full n x n shape, only this
    rank's columns populated) into the complete model -- on every rank, or (dst given)
    assembled on rank ``dst`` only; the other ranks take part in the collectives and
    return None."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    W_local = sp.csc_matrix(W_local)
    n = W_local.shape[1]
    if world == 1:
        return W_local
    dev = _default_device()
    counts = torch.from_numpy(np.diff(W_local.indptr).astype(np.int64)).to(dev)
    dist.all_reduce(counts, group=group)  # blocks are disjoint: the sum is the concatenation
    nnz_local = torch.tensor([W_local.nnz], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, nnz_local, group=group)
    sizes = [int(s) for s in sizes]
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
    # rank r's entries are the columns of its block in ascending column order, and blocks
    # ascend with the rank, so concatenation in rank order is the global CSC order --
    # provided every rank's populated columns form one contiguous block.
    indices = np.concatenate([t[:s].cpu().numpy() for t, s in zip(all_ind, sizes)])
    data = np.concatenate([t[:s].cpu().numpy() for t, s in zip(all_val, sizes)])
    indptr = np.concatenate([[0], np.cumsum(counts.cpu().numpy())])
    return sp.csc_matrix((data, indices, indptr), shape=(n, n))


def learn_sharded(mat, group=None, **opts):
    """SLIM_Learn over all ranks of ``group`` on a replicated DeviceMatrix.
    Returns (full W on every rank, this rank's stats, this rank's (begin, end))."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    blocks = partition_columns(mat.column_cost(), world)
    b, e = blocks[rank]
    W_local, stats = mat.learn(col_begin=b, col_end=e, **opts)
    W = gather_model(W_local, group) if world > 1 else W_local
    # the reductions of EstimateModelCD (estimate.c:371-373: error, objval) and the solve
    # counters, summed over the ranks; per-rank values stay under their own names
    stats = dict(stats)
    keys = [k for k in ("objval", "error", "nnzW", "G", "D", "U", "sweeps", "ncols_solved") if k in stats]
    if world > 1 and keys:
        import torch
        t = torch.tensor([float(stats[k]) for k in keys], dtype=torch.float64, device=_default_device())
        dist.all_reduce(t, group=group)
        totals = t.tolist()
    else:
        totals = [float(stats[k]) for k in keys]
    stats["totals"] = dict(zip(keys, totals))
    return W, stats, (b, e)
