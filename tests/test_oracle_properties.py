"""Properties of the CPU oracle that do not depend on any recorded output: what it returns is
the minimiser of the problem the reference states (README "min 1/2||r - Rx||^2 + l2/2||x||^2 +
l1||x||_1, x >= 0, x_i = 0"), checked through the optimality conditions on random matrices, and
its modes (visiting orders, fp64 3-pass vs fused fp32 arithmetic, per-item vs tile walk) agree
at a tight tolerance.  Together with tests/test_oracle_pins.py this is what makes the oracle
trustworthy as the checker of the GPU kernels."""
import numpy as np
import pytest
import scipy.sparse as sp

import slim_oracle as O


def _ratings(nu, ni, density, seed, binary=False):
    rng = np.random.default_rng(seed)
    R = sp.random(nu, ni, density=density, format="csr", random_state=rng, dtype=np.float32)
    R.data = np.ones(R.nnz, np.float32) if binary else rng.integers(1, 6, R.nnz).astype(np.float32)
    R.sort_indices()
    return R


@pytest.mark.parametrize("l1,l2,binary", [(1.0, 1.0, False), (0.1, 5.0, False), (3.0, 0.5, True),
                                          (0.0, 1.0, True)])
def test_oracle_solution_satisfies_kkt(l1, l2, binary):
    R = _ratings(400, 60, 0.08, 7, binary)
    W = O.learn_cd(R, l1r=l1, l2r=l2, optTol=1e-14, maxniters=200000, order=O.ORDER_PERM, seed=2,
                   aty=O.ATY_GRAM, nthreads=4).toarray().astype(np.float64)
    A = R.toarray().astype(np.float64)
    n = A.shape[1]
    assert W.min() >= 0 and np.abs(np.diag(W)).max() == 0
    resid = A - A @ W                        # column j: r_j - R x_j
    grad = A.T @ resid - l2 * W              # a_i.(y - Ax) - l2 x_i
    pos = W > 0
    assert np.abs(grad[pos] - l1).max() <= 5e-5          # active coordinates sit on the threshold
    off = ~pos
    off[np.arange(n), np.arange(n)] = False
    assert (grad[off] <= l1 + 5e-5).all()                # the others cannot improve the objective
    # (the oracle's filter drops |x| <= 1e-7, and fp32 output rounds: hence 5e-5 (observed 8e-6), not 1e-12)


def test_oracle_modes_agree_at_tight_tolerance():
    R = _ratings(600, 90, 0.06, 11)
    kw = dict(l1r=1.0, l2r=1.0, optTol=1e-14, maxniters=200000, aty=O.ATY_GRAM, nthreads=4)
    ref = O.learn_cd(R, order=O.ORDER_PERM, seed=1, **kw)
    for other in (O.learn_cd(R, order=O.ORDER_PERM, seed=99, **kw),            # another order
                  O.learn_cd(R, order=O.ORDER_NONE, **kw),                     # no shuffle at all
                  O.learn_cd(R, order=O.ORDER_LOCAL, seed=5, **kw),            # thread-local PRNG
                  O.learn_cd(R, order=O.ORDER_PERM, seed=1, fp32=True, **kw),  # fused fp32 arithmetic
                  O.learn_cd(R, order=O.ORDER_PERM, seed=1, l1r=1.0, l2r=1.0, optTol=1e-14,
                             maxniters=200000, aty=O.ATY_FULLSCAN, nthreads=4),
                  O.learn_cd_tile(R, tileP=32, seed=1, l1r=1.0, l2r=1.0, optTol=1e-14,
                                  maxniters=200000, nthreads=4),
                  O.learn_cd_tile(R, tileP=16, seed=3, l1r=1.0, l2r=1.0, optTol=1e-14,
                                  maxniters=200000, nthreads=4)):
        assert abs(ref - other).max() <= 2e-5


def test_oracle_topn_matches_dense_scores():
    """GetRecommendations (predict.c:15-71) against a dense restatement: scores = history @ W,
    history items excluded, best N by score."""
    R = _ratings(120, 50, 0.1, 3)
    W = O.learn_cd(R, order=O.ORDER_PERM, seed=1, aty=O.ATY_GRAM)
    ids, sc = O.predict(W, R, 5)
    S = (R @ sp.csr_matrix(W)).toarray()
    touched = ((R != 0).astype(np.float32) @ (sp.csr_matrix(W) != 0).astype(np.float32)).toarray() > 0
    for u in range(R.shape[0]):
        hist = set(R.indices[R.indptr[u]:R.indptr[u + 1]])
        cand = [(S[u, k], k) for k in range(R.shape[1]) if k not in hist and touched[u, k]]
        cand.sort(key=lambda t: -t[0])
        want = [k for _, k in cand[:5]]
        got = [k for k in ids[u] if k >= 0]
        assert len(got) == len(want)
        # same scores (ids may swap only where two scores tie to fp32 rounding)
        assert np.allclose(sorted(S[u, got], reverse=True), sorted(S[u, want], reverse=True), atol=1e-5)


def test_oracle_admm_against_a_numpy_restatement():
    """EstimateModelADMM (estimate.c:38-304) restated twice: the oracle's plain C loops and, here,
    numpy with LAPACK for the inverse and BLAS for the products -- two independent routes through
    the same recurrences must agree to rounding; plus what the iteration guarantees by
    construction (W >= 0, and a diagonal driven towards zero by gamma)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(2)
    R = sp.random(500, 120, density=0.08, random_state=rng, format="csr", dtype=np.float32)
    R.data = rng.integers(1, 6, R.nnz).astype(np.float32)
    l1, l2, rho = 0.7, 2.0, 10000.0
    W = O.learn_admm(R, l1r=l1, l2r=l2, nthreads=4).toarray().astype(np.float64)
    Rd = R.toarray().astype(np.float64)
    T = Rd.T @ Rd
    m = T.shape[0]
    P = np.linalg.inv(T + (l2 + rho) * np.eye(m))
    A = P @ T
    Wn = np.zeros((m, m))
    Cn = np.zeros((m, m))
    for _ in range(30):
        Wn = rho * Wn - Cn
        Tn = P @ Wn + A
        gamma = np.diag(Tn) / np.diag(P)
        B = Tn - P * gamma[None, :]
        alpha = B + Cn / rho
        Wn = np.maximum(np.maximum(alpha - l1 / rho, 0) - np.maximum(-alpha - l1 / rho, 0), 0)
        Cn = Cn + rho * (B - Wn)
    assert np.abs(W - Wn.astype(np.float32)).max() <= 1e-6
    assert W.min() >= 0 and np.abs(np.diag(W)).max() <= 1e-3
    assert (W > 0).sum() > m


def test_tile_walk_warm_start_reaches_the_per_item_warm_start(ml100k):
    """The tile walk's warm start (estimate.c:453-464 + cd.c:108-110 in oracle_learn_cd_tile_warm)
    against the per-item restatement's: from the same previous model both reach the same point at
    a tight tolerance, in fewer sweeps than a cold start; restarted from its own solution the
    walk needs one sweep."""
    R, _ = ml100k
    first = O.learn_cd(R, l1r=2.0, l2r=1.0, order=O.ORDER_PERM, seed=1, aty=O.ATY_GRAM, nthreads=8)
    kw = dict(l1r=1.0, l2r=0.5, optTol=1e-13, maxniters=100000, nthreads=8)
    a = O.learn_cd(R, order=O.ORDER_PERM, seed=2, aty=O.ATY_GRAM, imodel=first, **kw)
    b, sb, _, _ = O.learn_cd_tile(R, seed=2, imodel=first, return_stats=True, **kw)
    c, sc, _, _ = O.learn_cd_tile(R, seed=2, return_stats=True, **kw)
    d = abs(a - b)
    assert (float(d.max()) if d.nnz else 0.0) <= 2e-5
    assert sb["sweeps"].sum() < sc["sweeps"].sum()
    sol = O.learn_cd_tile(R, seed=1, nthreads=8)
    _, s2, _, _ = O.learn_cd_tile(R, seed=1, nthreads=8, imodel=sol, return_stats=True)
    assert s2["sweeps"].mean() <= 1.01


def test_time_budget_cuts_columns_off_and_reports_what_they_reached(ml100k):
    """bench.py's bounded all-cores sample (oracle_set_time_budget): past the budget no new sweep
    starts; a column cut off reports conv = -1 and the D it reached, a column never started
    conv = -2; the budget applies to ONE call."""
    R, _ = ml100k
    W, st, _, _ = O.learn_cd(R, order=O.ORDER_LOCAL, aty=O.ATY_GRAM, nthreads=2, return_stats=True, chunk=1)
    full = O.learn_seconds()
    assert (st["conv"] >= 0).all()
    O.set_time_budget(full / 4)
    _, st2, _, _ = O.learn_cd(R, order=O.ORDER_LOCAL, aty=O.ATY_GRAM, nthreads=2, return_stats=True, chunk=1)
    assert O.learn_seconds() < 0.8 * full
    assert (st2["conv"] == -2).sum() > 100 and (st2["conv"] >= 0).sum() > 50
    assert 0 < st2["D"].sum() < st["D"].sum()
    _, st3, _, _ = O.learn_cd(R, order=O.ORDER_LOCAL, aty=O.ATY_GRAM, nthreads=2, return_stats=True, chunk=1)
    assert (st3["conv"] >= 0).all() and abs(st3["D"].sum() - st["D"].sum()) <= 0.01 * st["D"].sum()


def test_item_space_restatement_walks_to_the_same_model():
    """The algebra the item-space kernel (cd_gram.hpp) rests on, checked on the CPU in fp64 and
    independently of any GPU code: carrying g_i = a_i.r = aTy_i - sum_j G_ij x_j over the ITEMS
    (G = R^T R, an update is g -= d G[i, :]) and visiting num = g_i + x_i G_ii is the
    coordinate descent of cd.c:112-139 -- a plain numpy walk of that form, in the tile's visiting
    order (union of the 32 active sets, keyed permutation, per-problem cap and stop rule,
    epsilon rule of cd.c:27), reaches the oracle's tile walk (user space, three passes per visit)
    to the last float of the model, with the same sweep counts."""
    rng = np.random.default_rng(3)
    R = sp.random(400, 90, density=0.12, format="csr", random_state=rng, dtype=np.float32)
    R.data = rng.integers(1, 6, R.nnz).astype(np.float32)
    R.sort_indices()
    l1, l2, tol, seed, eps = 0.5, 1.5, 1e-9, 5, 1e-7
    order = O.tile_work_order(R)
    Wo, so, _, _ = O.learn_cd_tile(R, tileP=32, order=order, l1r=l1, l2r=l2, optTol=tol, seed=seed,
                                   return_stats=True)
    L = O.lib()
    A = R.toarray().astype(np.float64)
    G = A.T @ A
    ncols = R.shape[1]
    nnz_col = np.diff(R.tocsc().indptr)
    cn = np.sqrt(np.diag(G).astype(np.float32)).astype(np.float32).astype(np.float64)   # setup.c:130
    W = np.zeros((ncols, ncols), np.float32)
    sweeps = np.zeros(ncols, np.int32)
    for g in range((order.size + 31) // 32):
        members = order[g * 32:(g + 1) * 32]
        aty = {int(iC): G[:, iC].astype(np.float32).astype(np.float64) for iC in members}   # float key
        act = {iC: (aty[iC] > l1) & (np.arange(ncols) != iC) for iC in aty}
        union = np.flatnonzero(np.any([act[iC] for iC in aty], axis=0))
        nu = union.size
        for iC in aty:
            x = np.zeros(ncols)
            gvec = aty[iC].copy()            # g = aTy - G xeff, xeff = 0
            maxit = min(50 * int(nnz_col[iC]), 10000)
            t = 0
            while t < maxit:
                key = L.oracle_perm_key(seed, g, t)
                dlt = 0.0
                for p in range(nu):
                    i = int(union[L.oracle_perm_index(p, nu, key)])
                    if not act[iC][i]:
                        continue
                    xi = x[i]
                    xeff = xi if abs(xi) > eps else 0.0
                    num = gvec[i] + xeff * G[i, i]
                    nx = (num - l1) / (cn[i] * cn[i] + l2) if num > l1 else 0.0
                    neff = nx if abs(nx) > eps else 0.0
                    if neff != xeff:
                        gvec -= (neff - xeff) * G[i, :]
                    x[i] = nx
                    dlt += (nx - xi) ** 2
                t += 1
                if dlt < tol:
                    break
            sweeps[iC] = t if (t < maxit or dlt < tol) else maxit + 1
            keep = np.abs(x) > eps
            W[keep, iC] = x[keep].astype(np.float32)
    d = abs(sp.csc_matrix(W) - Wo)
    assert Wo.nnz > 500 and (d.max() if d.nnz else 0.0) <= 2e-7
    assert (sweeps == so["sweeps"]).mean() >= 0.98


def test_item_space_grid_step_from_the_carried_g_is_the_warm_started_walk():
    """The algebra of the carried g (cd_gramr.hpp: g_save / g_load), on the CPU in fp64
    and independently of any GPU code.  A grid step that only moves l2 keeps every active set (the
    screen aTy > l1 does not see l2), so the warm start of the reference -- x from the previous model,
    y-hat = A x folded in before the first sweep (estimate.c:453-464, cd.c:108-110) -- is, in item space,
    g = aTy - sum_j x_j G[j, :] over the kept coefficients: exactly the g the previous walk ends with
    (the epsilon rule of cd.c:27 governs what enters g, what is folded and what is kept alike).  A numpy
    walk that carries g and the full x across the step reaches the oracle's warm-started tile walk
    (user space; ITS previous model) to the last float, with the same sweep counts; and its carried g
    equals the recomputed fold."""
    rng = np.random.default_rng(11)
    R = sp.random(500, 64, density=0.15, format="csr", random_state=rng, dtype=np.float32)
    R.data[:] = 1.0
    R.sort_indices()
    l1, tol, seed, eps = 0.5, 1e-10, 3, 1e-7
    order = O.tile_work_order(R)
    L = O.lib()
    A = R.toarray().astype(np.float64)
    G = A.T @ A
    ncols = R.shape[1]
    nnz_col = np.diff(R.tocsc().indptr)
    cn = np.sqrt(np.diag(G).astype(np.float32)).astype(np.float32).astype(np.float64)

    def walk(l2, state):
        """one pair over every tile; state[iC] = (x, g) of the previous pair (None: cold)"""
        W = np.zeros((ncols, ncols), np.float32)
        sweeps = np.zeros(ncols, np.int32)
        out = {}
        for grp in range((order.size + 31) // 32):
            members = [int(c) for c in order[grp * 32:(grp + 1) * 32]]
            aty = {iC: G[:, iC].astype(np.float32).astype(np.float64) for iC in members}
            act = {iC: (aty[iC] > l1) & (np.arange(ncols) != iC) for iC in members}
            union = np.flatnonzero(np.any([act[iC] for iC in members], axis=0))
            nu = union.size
            for iC in members:
                if state is None:
                    x, gvec = np.zeros(ncols), aty[iC].copy()
                else:
                    xp, gp = state[iC]
                    # what SLIM_Learn's warm start would form from the MODEL of the previous pair
                    xm = np.where(np.abs(xp) > eps, xp.astype(np.float32).astype(np.float64), 0.0)
                    xm = np.where(act[iC], np.maximum(xm, 0.0), 0.0)
                    fold = aty[iC] - G @ xm
                    x, gvec = xm.copy(), gp.copy()      # the carried g in place of the fold ...
                    assert np.abs(gvec - fold).max() <= 1e-6 * max(1.0, np.abs(fold).max())   # ... which it equals
                maxit = min(50 * int(nnz_col[iC]), 10000)
                t, dlt = 0, 0.0
                while t < maxit:
                    key = L.oracle_perm_key(seed, grp, t)
                    dlt = 0.0
                    for p in range(nu):
                        i = int(union[L.oracle_perm_index(p, nu, key)])
                        if not act[iC][i]:
                            continue
                        xi = x[i]
                        xeff = xi if abs(xi) > eps else 0.0
                        num = gvec[i] + xeff * G[i, i]
                        nx = (num - l1) / (cn[i] * cn[i] + l2) if num > l1 else 0.0
                        neff = nx if abs(nx) > eps else 0.0
                        if neff != xeff:
                            gvec -= (neff - xeff) * G[i, :]
                        x[i] = nx
                        dlt += (nx - xi) ** 2
                    t += 1
                    if dlt < tol:
                        break
                sweeps[iC] = t if (t < maxit or dlt < tol) else maxit + 1
                keep = np.abs(x) > eps
                W[keep, iC] = x[keep].astype(np.float32)
                out[iC] = (x, gvec)
        return sp.csc_matrix(W), sweeps, out

    W1, s1, st1 = walk(1.0, None)
    W2, s2, _ = walk(4.0, st1)
    Wo1, so1, _, _ = O.learn_cd_tile(R, tileP=32, order=order, l1r=l1, l2r=1.0, optTol=tol, seed=seed,
                                     binary=True, return_stats=True)
    Wo2, so2, _, _ = O.learn_cd_tile(R, tileP=32, order=order, l1r=l1, l2r=4.0, optTol=tol, seed=seed,
                                     binary=True, return_stats=True, imodel=Wo1)
    for Wn, Wo, sn, so in ((W1, Wo1, s1, so1), (W2, Wo2, s2, so2)):
        d = abs(Wn - Wo)
        assert Wo.nnz > 300 and (d.max() if d.nnz else 0.0) <= 3e-7
        assert (sn == so["sweeps"]).mean() >= 0.97
