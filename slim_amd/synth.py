"""Synthetic user x item rating matrices for the large benchmark configurations
(BASELINE.json configs[3] and [4]): 1M users x 100K items / ~1e9 nnz
and 10M x 20K / ~1e9 nnz are too big to ship or to build on the host in reasonable
time, so they are generated where they are used -- on the GPU -- from a seed.

Model: item popularity ~ 1/(rank + c)^0.8, user activity
log-normal clipped to [5, 5000] and rescaled to the requested nnz, items drawn from
the popularity law per user without replacement (oversample, merge duplicates, keep a
random subset of the wanted size), values either all 1.0 (implicit feedback) or ratings
1..5 with P = (.05, .05, .10, .30, .50).  Output: CSR with ascending item ids inside
each row, int64 rowptr, int32 rowind, float32 rowval.

torch is used as the array library (same code on "cpu" for tests and on "cuda");
nothing here is on the training path.
"""
import math

import torch


def _activity(nrows, target_nnz, gen, device, lo=5, hi=5000):
    z = torch.randn(nrows, generator=gen, device=device, dtype=torch.float32)
    act = torch.exp(1.0 * z)
    hi = min(hi, 10 ** 9)
    scale = target_nnz / float(act.sum())
    deg = torch.clamp((act * scale).round(), lo, hi)
    for _ in range(8):  # clipping moves the total: rescale the unclipped part a few times
        total = float(deg.sum())
        if abs(total - target_nnz) <= 1e-3 * target_nnz:
            break
        scale *= target_nnz / total
        deg = torch.clamp((act * scale).round(), lo, hi)
    return deg.to(torch.int64)


def generate_csr(nrows, ncols, target_nnz, seed=1, ratings=False, device="cpu",
                 chunk_nnz=1 << 26, pop_exponent=0.8, pop_offset=10.0):
    """Returns (rowptr int64[nrows+1], rowind int32[nnz], rowval float32[nnz]) on ``device``."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    hi = min(5000, max(5, ncols // 2))
    deg = _activity(nrows, target_nnz, gen, dev, lo=min(5, ncols), hi=hi)
    # popularity CDF over items; item ids are a random relabelling of the popularity rank
    rank = torch.arange(ncols, device=dev, dtype=torch.float64)
    w = 1.0 / torch.pow(rank + pop_offset, pop_exponent)
    cdf = torch.cumsum(w / w.sum(), 0).to(torch.float32)
    cdf[-1] = 1.0
    relabel = torch.randperm(ncols, generator=gen, device=dev)

    # Items are drawn WITHOUT replacement per user: every user draws
    # ~1.5x its degree from the popularity law, duplicates are merged, and a uniformly random
    # subset of exactly deg[u] of the distinct items is kept (random, not "the first by id":
    # that would tie popularity to the id).  A user whose oversampled draw still holds fewer
    # than deg[u] distinct items keeps them all (rare; the realised nnz is reported).
    draws = (deg * 3 + 1) // 2 + 16
    draws = torch.minimum(draws, torch.clamp(deg * 8, min=16))
    dcum = torch.cumsum(draws, 0)
    dstart = dcum - draws
    ind_parts, cnt_parts = [], []
    r0 = 0
    while r0 < nrows:
        # rows [r0, r1) whose draws fit one chunk
        base = int(dstart[r0])
        r1 = int(torch.searchsorted(dcum, torch.tensor([base + chunk_nnz], device=dev),
                                    right=True)[0])
        r1 = max(r1, r0 + 1)
        r1 = min(r1, nrows)
        d = draws[r0:r1]
        n = int(d.sum())
        users = torch.repeat_interleave(torch.arange(r0, r1, device=dev, dtype=torch.int64), d)
        u = torch.rand(n, generator=gen, device=dev, dtype=torch.float32)
        items = relabel[torch.searchsorted(cdf, u).clamp_(max=ncols - 1)]
        key = users * ncols + items
        del users, u, items
        key = torch.unique(key)  # sorted: by user, then item; duplicates merged
        lu = (key // ncols) - r0  # local user of every distinct pair
        have = torch.bincount(lu, minlength=r1 - r0)
        first = torch.cumsum(have, 0) - have
        # random rank of every pair inside its user: sort by (user, random tag)
        tag = torch.randint(0, 1 << 31, (key.numel(),), generator=gen, device=dev, dtype=torch.int64)
        pos = torch.argsort(lu * (1 << 31) + tag)
        rank = torch.empty_like(pos)
        rank[pos] = torch.arange(key.numel(), device=dev, dtype=torch.int64)
        keep = (rank - first[lu]) < deg[r0:r1][lu]
        del tag, pos, rank
        key = key[keep]  # still sorted by (user, item)
        ind_parts.append((key % ncols).to(torch.int32))
        cnt_parts.append(torch.bincount((key // ncols) - r0, minlength=r1 - r0))
        del key, lu, keep
        r0 = r1
    rowind = torch.cat(ind_parts)
    counts = torch.cat(cnt_parts)
    rowptr = torch.zeros(nrows + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0)
    nnz = int(rowptr[-1])
    if ratings:
        probs = torch.tensor([0.05, 0.05, 0.10, 0.30, 0.50], device=dev)
        rowval = (torch.multinomial(probs, nnz, replacement=True, generator=gen) + 1).to(torch.float32) \
            if nnz < (1 << 24) else _ratings_big(nnz, gen, dev)
    else:
        rowval = torch.ones(nnz, dtype=torch.float32, device=dev)
    assert rowind.numel() == nnz
    return rowptr, rowind, rowval


def _ratings_big(nnz, gen, dev):
    edges = torch.tensor([0.05, 0.10, 0.20, 0.50], device=dev)
    out = torch.empty(nnz, dtype=torch.float32, device=dev)
    step = 1 << 26
    for s in range(0, nnz, step):
        e = min(nnz, s + step)
        u = torch.rand(e - s, generator=gen, device=dev)
        out[s:e] = (torch.bucketize(u, edges) + 1).to(torch.float32)
    return out


CONFIGS = {
    # name: (nrows, ncols, target nnz) -- BASELINE.json configs[3], its 0.1 % variant, configs[4]
    "c4": (1_000_000, 100_000, 1_000_000_000),
    "c4-0.1pct": (1_000_000, 100_000, 100_000_000),
    "c5": (10_000_000, 20_000, 1_000_000_000),
}


def scaled(name, scale):
    """A configuration shrunk by ``scale`` in both dimensions (density kept)."""
    nr, nc, nz = CONFIGS[name]
    s = float(scale)
    return max(64, int(nr * s)), max(64, int(nc * s)), max(1024, int(nz * s * s))
