"""Sharded solves: the shard options of SLIMGPU_Learn, SLIM_Learn over several GPUs inside one
process (csrc/multi_gpu.cpp; reference: the OpenMP team inside SLIM_Learn, api.c:69-85 ->
estimate.c:371-373,402), the one-process-per-GPU driver on a device, and the cluster fallback.

The GPU box of the test tier has ONE device: the multi-device code runs there with
SLIM_GPU_DEVICES=0,0 (two host threads, two replicas of R, two shards, one GPU); with two or more
devices present the same tests also run on distinct devices."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN, ROOT
from slim_amd import _lib
from slim_amd.constants import SLIM_NOPTIONS, Opt
from slim_amd.engine import KERNEL_GRAM, KERNEL_TILE, KERNEL_WAVE_LDS, DeviceMatrix, model_to_scipy

pytestmark = pytest.mark.gpu


def maxdiff(a, b):
    d = abs(a - b)
    return float(d.max()) if d.nnz else 0.0


def _slim_learn(R, ngpus=None, dbglvl=0, **dopts):
    lib = _lib.load()
    io = np.full(SLIM_NOPTIONS, -1, np.int32)
    do = np.full(SLIM_NOPTIONS, -1.0, np.float64)
    io[Opt.DBGLVL] = dbglvl
    if ngpus is not None:
        io[Opt.GPU_NGPUS] = ngpus
    for k, v in dopts.items():
        do[getattr(Opt, k)] = v
    st = C.c_int32(0)
    h = lib.SLIM_Learn(R.shape[0], R.indptr.astype(np.intp), R.indices.astype(np.int32),
                       R.data.astype(np.float32).ctypes.data_as(C.c_void_p),
                       io.ctypes.data_as(C.c_void_p), do.ctypes.data_as(C.c_void_p), None,
                       C.byref(st))
    if not h:
        return None, st.value, _lib.last_error()
    stats = _lib.Stats()
    lib.SLIMGPU_LastStats(C.byref(stats))
    return model_to_scipy(lib, h), st.value, stats.as_dict()


@pytest.mark.parametrize("kernel,geom", [(KERNEL_WAVE_LDS, {}), (KERNEL_TILE, {"cluster": 2}), (KERNEL_GRAM, {})])
def test_shards_do_not_change_a_column(ml100k, kernel, geom):
    """Shard i of c = granules i, i + c, ... of the cost-ordered work list, visiting order keyed
    by the tile's position in the unsharded list: the union of the shards IS the single solve."""
    R, _ = ml100k
    m = DeviceMatrix.from_scipy(R)
    W, st = m.learn(seed=1, kernel=kernel, **geom)
    for count in (2, 3):
        parts = [m.learn(seed=1, kernel=kernel, shard=(i, count), **geom) for i in range(count)]
        assert sum(p[1]["ncols_solved"] for p in parts) == R.shape[1]
        total = parts[0][0]
        for p in parts[1:]:
            assert total.multiply(p[0]).nnz == 0          # column-disjoint
            total = total + p[0]
        assert maxdiff(total, W) == 0.0 and total.nnz == W.nnz
        assert abs(sum(p[1]["objval"] for p in parts) - st["objval"]) <= 1e-6 * st["objval"]
    with pytest.raises(RuntimeError):
        m.learn(seed=1, shard=(2, 2))
    m.close()


def test_slim_learn_over_two_shards_equals_one_gpu(ml100k, monkeypatch):
    """The C ABI: SLIM_Learn with option slot 19 (ngpus) = 2 returns the model of ngpus = 1, bit
    for bit, and the reductions (estimate.c:371-373) are summed over the team."""
    R, _ = ml100k
    W1, st1, s1 = _slim_learn(R, L1R=1.0, L2R=1.0)
    assert st1 == 1
    ndev = _lib.load().SLIMGPU_DeviceCount()
    if ndev < 2:
        monkeypatch.setenv("SLIM_GPU_DEVICES", "0,0")
    W2, st2, s2 = _slim_learn(R, ngpus=2, L1R=1.0, L2R=1.0)
    assert st2 == 1, s2
    assert W2.nnz == W1.nnz and maxdiff(W1, W2) == 0.0
    assert s2["ncols_solved"] == R.shape[1] and s2["nnzW"] == s1["nnzW"]
    assert abs(s2["objval"] - s1["objval"]) <= 1e-6 * s1["objval"]
    assert s2["G"] == s1["G"] and s2["D"] == s1["D"]
    # SLIM_GPU_NGPUS reaches callers that cannot set the slot (the reference's unchanged
    # Python wrapper and CLIs)
    monkeypatch.setenv("SLIM_GPU_NGPUS", "2")
    W3, st3, _ = _slim_learn(R, L1R=1.0, L2R=1.0)
    assert st3 == 1 and maxdiff(W1, W3) == 0.0


def test_slim_learn_item_space_over_two_replicas(monkeypatch):
    """The in-library team on the item-space path: SLIM_Learn with ngpus = 2 and option slot 15 =
    SLIMGPU_KERNEL_GRAM -- every replica of R builds its own G = R^T R and solves its shard --
    returns the ngpus = 1 model bit for bit (a column's walk does not depend on the shards)."""
    rng = np.random.default_rng(5)
    R = sp.random(40000, 3000, density=0.004, format="csr", random_state=rng, dtype=np.float32)
    R.data = rng.integers(1, 6, R.nnz).astype(np.float32)
    R.sort_indices()
    lib = _lib.load()

    def learn(ngpus):
        io = np.full(SLIM_NOPTIONS, -1, np.int32)
        do = np.full(SLIM_NOPTIONS, -1.0, np.float64)
        io[Opt.GPU_KERNEL] = KERNEL_GRAM
        io[Opt.GPU_NGPUS] = ngpus
        do[Opt.L1R], do[Opt.L2R] = 1.0, 0.5
        st = C.c_int32(0)
        h = lib.SLIM_Learn(R.shape[0], R.indptr.astype(np.intp), R.indices.astype(np.int32),
                           R.data.astype(np.float32).ctypes.data_as(C.c_void_p),
                           io.ctypes.data_as(C.c_void_p), do.ctypes.data_as(C.c_void_p), None, C.byref(st))
        assert h, _lib.last_error()
        stats = _lib.Stats()
        lib.SLIMGPU_LastStats(C.byref(stats))
        return model_to_scipy(lib, h), stats.as_dict()

    W1, s1 = learn(1)
    assert s1["kernel"] == KERNEL_GRAM and W1.nnz > 100000
    if lib.SLIMGPU_DeviceCount() < 2:
        monkeypatch.setenv("SLIM_GPU_DEVICES", "0,0")
    W2, s2 = learn(2)
    assert s2["kernel"] == KERNEL_GRAM and W2.nnz == W1.nnz and maxdiff(W1, W2) == 0.0


def test_slim_learn_more_gpus_than_devices_is_an_input_error(ml100k, monkeypatch):
    R, _ = ml100k
    monkeypatch.delenv("SLIM_GPU_DEVICES", raising=False)
    ndev = _lib.load().SLIMGPU_DeviceCount()
    W, st, msg = _slim_learn(R, ngpus=ndev + 1)
    assert W is None and st == -2 and "ngpus" in msg


def test_rccl_staging_path(ml100k, monkeypatch):
    """SLIM_GPU_STAGE=rccl: R reaches the devices through ncclBroadcast (librccl loaded on
    demand) instead of one H2D copy per device; with one device the broadcast is the identity."""
    R, _ = ml100k
    W1, _, _ = _slim_learn(R, L1R=2.0, L2R=0.5)
    monkeypatch.setenv("SLIM_GPU_STAGE", "rccl")
    ndev = _lib.load().SLIMGPU_DeviceCount()
    W2, st, msg = _slim_learn(R, ngpus=min(ndev, 2), L1R=2.0, L2R=0.5)
    assert st == 1, msg
    assert maxdiff(W1, W2) == 0.0


def test_view_staging_path(ml100k, monkeypatch, capfd):
    """SLIM_GPU_STAGE=view: one device stages R, the other receives the finished CSR + CSC +
    column scalars device to device (no second sort); the two-shard model is the one-GPU model.
    On a one-GPU box both replicas live on device 0 (the copy is then a D2D copy)."""
    R, _ = ml100k
    W1, st1, _ = _slim_learn(R, L1R=1.0, L2R=1.0)
    assert st1 == 1
    if _lib.load().SLIMGPU_DeviceCount() < 2:
        monkeypatch.setenv("SLIM_GPU_DEVICES", "0,0")
    monkeypatch.setenv("SLIM_GPU_STAGE", "view")
    monkeypatch.setenv("SLIM_GPU_TRACE", "1")
    W2, st2, s2 = _slim_learn(R, ngpus=2, L1R=1.0, L2R=1.0)
    assert st2 == 1, s2
    assert "staging 'view'" in capfd.readouterr().err
    assert W2.nnz == W1.nnz and maxdiff(W1, W2) == 0.0
    # devices that cannot reach each other (hipDeviceCanAccessPeer says no; SLIM_GPU_PEER=0 plays
    # that answer): the finished views travel through a pinned host buffer instead
    monkeypatch.setenv("SLIM_GPU_PEER", "0")
    W3, st3, s3 = _slim_learn(R, ngpus=2, L1R=1.0, L2R=1.0)
    assert st3 == 1, s3
    assert W3.nnz == W1.nnz and maxdiff(W1, W3) == 0.0


def test_explicit_device_wins_over_the_environment(ml100k, monkeypatch):
    """a one-process-per-GPU rank that inherits SLIM_GPU_DEVICES must not replicate R
    on every listed device -- an explicit device option with ngpus <= 1 is honoured, and a list
    longer than ngpus is cut to ngpus entries."""
    R, _ = ml100k
    monkeypatch.setenv("SLIM_GPU_DEVICES", "0,0,0")
    m = DeviceMatrix.from_scipy(R, device=0)
    W, st = m.learn(seed=1)
    assert st["ncols_solved"] == R.shape[1]      # one matrix, one solve: no fan-out
    m.close()
    W1, st1, s1 = _slim_learn(R, L1R=1.0, L2R=1.0)    # ngpus unset: the first listed device only
    assert st1 == 1 and s1["ncols_solved"] == R.shape[1]


def test_mselect_grid_over_two_shards(automotive_triplets, monkeypatch, capsys):
    """Py_SLIM_Mselect keeps R on every device of the team across the grid."""
    from slim_amd.interface import SLIM, SLIMatrix
    trn, tst = automotive_triplets
    if _lib.load().SLIMGPU_DeviceCount() < 2:
        monkeypatch.setenv("SLIM_GPU_DEVICES", "0,0")
    monkeypatch.setenv("SLIM_GPU_NGPUS", "2")
    trainmat = SLIMatrix(trn)
    valmat = SLIMatrix(tst, trainmat)
    model = SLIM()
    params = {"dbglvl": 0, "algo": "cd", "nthreads": 1, "optTol": 1e-7, "niters": 100}
    model.mselect(params, trainmat, valmat, [10, 20], [0.1, 50], nrcmds=10)
    l1, l2, hr, ar = model.mselect_result["bestHR"]
    assert (l1, l2) == (20.0, 0.1) and abs(hr - 0.1404) <= 5e-4     # UserGuide.ipynb:276


@pytest.mark.timeout(120, method="thread")
def test_cluster_timeout_falls_back_to_unclustered_solve(ml100k, monkeypatch, capfd):
    """A cluster whose member never becomes resident (CU mask, second tenant) times out; the
    launch is void and everything is solved again without clusters -- a clean result, not an
    error.  SLIM_GPU_TEST_DROP_MEMBER launches the last cluster one workgroup short."""
    R, _ = ml100k
    m = DeviceMatrix.from_scipy(R)
    want, _ = m.learn(seed=1, kernel=KERNEL_TILE, cluster=1)
    monkeypatch.setenv("SLIM_GPU_TEST_HOOKS", "1")
    monkeypatch.setenv("SLIM_GPU_TEST_DROP_MEMBER", "1")
    got, st = m.learn(seed=1, kernel=KERNEL_TILE, cluster=4)
    assert "re-solving" in capfd.readouterr().err
    assert got.nnz == want.nnz and maxdiff(got, want) == 0.0
    m.close()


@pytest.mark.timeout(180, method="thread")
def test_cluster_timeout_fallback_with_a_large_user_range(monkeypatch, capfd):
    """The same fallback on 200 000 users: without clusters a member's user range --
    and with it the LDS bitmap of the screen pass -- is four times the clustered one (25 KB
    against 6 KB here, several LDS allocation granules apart); the relaunch must size its
    dynamic LDS for the new geometry or the screen pass silently drops users."""
    rng = np.random.default_rng(23)
    R = sp.random(200000, 64, density=0.02, format="csr", random_state=rng, dtype=np.float32)
    R.data[:] = 1.0 + np.floor(rng.random(R.nnz) * 5).astype(np.float32)
    m = DeviceMatrix.from_scipy(R)
    want, _ = m.learn(seed=1, kernel=KERNEL_TILE, cluster=1)
    monkeypatch.setenv("SLIM_GPU_TEST_HOOKS", "1")
    monkeypatch.setenv("SLIM_GPU_TEST_DROP_MEMBER", "1")
    got, st = m.learn(seed=1, kernel=KERNEL_TILE, cluster=4)
    assert "re-solving" in capfd.readouterr().err
    assert want.nnz > 0 and got.nnz == want.nnz and maxdiff(got, want) == 0.0
    m.close()


def test_test_hooks_are_inert_without_the_master_switch(ml100k, monkeypatch, capfd):
    """SLIM_GPU_TEST_DROP_MEMBER inherited from somebody's environment must not void launches:
    the hooks act only together with SLIM_GPU_TEST_HOOKS=1."""
    R, _ = ml100k
    m = DeviceMatrix.from_scipy(R)
    monkeypatch.delenv("SLIM_GPU_TEST_HOOKS", raising=False)
    monkeypatch.setenv("SLIM_GPU_TEST_DROP_MEMBER", "1")
    got, st = m.learn(seed=1, kernel=KERNEL_TILE, cluster=4, col_begin=0, col_end=128)
    assert "re-solving" not in capfd.readouterr().err and got.nnz > 0
    m.close()


@pytest.mark.timeout(240, method="thread")
def test_clusters_under_a_cu_mask(tmp_path):
    """Half of the compute units masked off (ROC_GLOBAL_CU_MASK): either the runtime reports the
    smaller device and the grid shrinks with it, or the clusters time out and the fallback
    takes over; both must end with the unmasked result."""
    code = r"""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, %r)
from slim_amd.engine import DeviceMatrix, KERNEL_TILE
from slim_amd.io import read_csr_text
R = read_csr_text(%r)
m = DeviceMatrix.from_scipy(R)
W, st = m.learn(seed=1, kernel=KERNEL_TILE, cluster=8)
sp.save_npz(sys.argv[1], sp.csc_matrix(W))
""" % (ROOT, os.path.join(GOLDEN, "ml100k-train.csr"))
    outs = []
    for name, mask in (("full", None), ("masked", "0x" + "f" * 32)):
        env = dict(os.environ)
        env.pop("ROC_GLOBAL_CU_MASK", None)
        if mask:
            env["ROC_GLOBAL_CU_MASK"] = mask
            env["HSA_CU_MASK"] = "0:0-127"
        path = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True,
                           text=True, timeout=200)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(sp.load_npz(path))
    assert outs[0].nnz == outs[1].nnz and maxdiff(outs[0], outs[1]) <= 2e-5


def test_cli_ngpus(tmp_path, monkeypatch):
    """slim_learn -ngpus=2 writes the model slim_learn writes."""
    exe = os.path.join(ROOT, "slim_amd", "bin", "slim_learn")
    trn = os.path.join(GOLDEN, "ml100k-train.csr")
    env = dict(os.environ)
    a, b = str(tmp_path / "a.model"), str(tmp_path / "b.model")
    subprocess.run([exe, "-l1r=1", "-l2r=1", trn, a], check=True, capture_output=True, env=env)
    if _lib.load().SLIMGPU_DeviceCount() < 2:
        env["SLIM_GPU_DEVICES"] = "0,0"
    r = subprocess.run([exe, "-l1r=1", "-l2r=1", "-ngpus=2", trn, b], check=True,
                       capture_output=True, env=env, text=True)
    assert open(a).read() == open(b).read()


# ---- one process per GPU (slim_amd/distributed.py) on a device -------------------------------
def _rank_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from slim_amd.distributed import broadcast_csr, learn_sharded
    from slim_amd.io import read_csr_text
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", rank % ndev)
    torch.cuda.set_device(dev)
    # RCCL needs one device per rank; with fewer devices than ranks the transport is gloo
    # (host tensors) while the solve still runs on the GPU
    backend = "nccl" if ndev >= world else "gloo"
    dist.init_process_group(backend, rank=rank, world_size=world,
                            **({"device_id": dev} if backend == "nccl" else {}))
    try:
        where = dev if backend == "nccl" else torch.device("cpu")
        if rank == 0:
            R = read_csr_text(os.path.join(GOLDEN, "ml100k-train.csr"))
            ptr = torch.from_numpy(R.indptr.astype(np.int64)).to(where)
            ind = torch.from_numpy(R.indices.astype(np.int32)).to(where)
            val = torch.from_numpy(R.data.astype(np.float32)).to(where)
        else:
            ptr = ind = val = None
        ptr, ind, val = broadcast_csr(ptr, ind, val, src=0)
        ptr, ind, val = ptr.to(dev), ind.to(dev), val.to(dev)
        torch.cuda.synchronize()
        mat = DeviceMatrix.from_device_ptrs(ptr.numel() - 1, 0, ptr.data_ptr(), ind.data_ptr(),
                                            val.data_ptr(), keepalive=(ptr, ind, val),
                                            device=dev.index)
        for partition in ("shards", "blocks"):
            W, stats, _ = learn_sharded(mat, seed=1, partition=partition)
            sp.save_npz(os.path.join(out_dir, "%s%d.npz" % (partition, rank)), sp.csc_matrix(W))
            np.save(os.path.join(out_dir, "%s%d.npy" % (partition, rank)),
                    np.array([stats["totals"]["ncols_solved"], stats["totals"]["nnzW"]]))
        # G = R^T R formed once by the ranks together (row blocks, one broadcast per block), then the
        # sharded solve in item space on it
        from slim_amd.distributed import build_gram_sharded
        from slim_amd.engine import KERNEL_GRAM
        build_gram_sharded(mat)
        G = mat.gram_rows_tensor(0, mat.ncols)[:, :mat.ncols].cpu().numpy()
        np.save(os.path.join(out_dir, "G%d.npy" % rank), G)
        W, stats, _ = learn_sharded(mat, seed=1, kernel=KERNEL_GRAM)
        assert stats["gram_build_ms"] == 0
        sp.save_npz(os.path.join(out_dir, "gram%d.npz" % rank), sp.csc_matrix(W))
        mat.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300, method="thread")
def test_one_process_per_gpu_driver_on_device(ml100k, tmp_path):
    """learn_sharded with DeviceMatrix.from_device_ptrs in two ranks (RCCL when two devices are
    present, gloo transport otherwise): every rank ends with the single-GPU model."""
    import socket
    import torch.multiprocessing as mp
    R, _ = ml100k
    m = DeviceMatrix.from_scipy(R)
    want, _ = m.learn(seed=1)
    m.close()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for partition in ("shards", "blocks"):
        w0 = sp.load_npz(str(tmp_path / (partition + "0.npz")))
        w1 = sp.load_npz(str(tmp_path / (partition + "1.npz")))
        assert abs(w0 - w1).nnz == 0 and maxdiff(w0, want) == 0.0 and w0.nnz == want.nnz
        t = np.load(str(tmp_path / (partition + "0.npy")))
        assert t[0] == R.shape[1] and t[1] == want.nnz
    # G formed by the two ranks together is R^T R on both, and the item-space model on it is the
    # single-GPU item-space model
    m = DeviceMatrix.from_scipy(R)
    want_g, _ = m.learn(seed=1, kernel=5)
    m.close()
    G = (R.T @ R).toarray().astype(np.float32)
    for rank in (0, 1):
        assert np.array_equal(np.load(str(tmp_path / ("G%d.npy" % rank))), G)
        assert maxdiff(sp.load_npz(str(tmp_path / ("gram%d.npz" % rank))), want_g) == 0.0


@pytest.mark.timeout(420, method="thread")
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_multi_rank_branch(scaling):
    """bench.py's own world > 1 path (shards per rank, barrier + max-over-ranks timing, gather of
    the learned columns on rank 0), launched the way the driver launches it.  On a one-GPU box
    the two ranks share the device and the collectives run over gloo; with two devices, RCCL."""
    import json
    import socket
    import torch
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "ml100k",
           "--backend", backend, "--scaling", scaling, "--cpu-seconds", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == scaling
    assert out["value"] > 0 and out["unit"] == "item-columns/s"
    assert out["config"]["columns_per_step"] == 1683
    assert out["roofline"]["nnzW"] > 0


@pytest.mark.timeout(420, method="thread")
def test_bench_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the
    one-rank-per-GPU job itself.  On a one-GPU box the two ranks share the device over gloo."""
    import json
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--workload", "ml100k", "--backend", backend, "--cpu-seconds", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert out["ranks"]["world"] == 2 and len(out["ranks"]["kernel_ms_per_rank"]) == 2
    assert all(v > 0 for v in out["ranks"]["kernel_ms_per_rank"])


@pytest.mark.timeout(900, method="thread")
def test_bench_four_ranks_whole_matrix_extras():
    """The N >= 4 branch of bench.py as the driver will run it at N = 4 / 8, on a 1/50-scale C4
    (20 000 x 2 000): after the timed steps one step over ALL item columns sharded over the ranks
    (`strong_whole_matrix`) and the same step in item space (`item_space`: G = R^T R formed in row
    blocks by the ranks together, --shard-gram, then every rank solves its shard) -- both agreed on by all ranks before any collective, so a
    failure in an extra step costs its key, not the line.  On a one-GPU box the four ranks share
    the device over gloo (clusters of 1: co-residency across processes is not a given there)."""
    import json
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 4 else "gloo"
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1",
           "--workload", "c4", "--scale", "0.02", "--batch", "256", "--cluster", "1",
           "--backend", backend, "--cpu-seconds", "0", "--shard-gram"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["value"] > 0 and len(out["ranks"]["kernel_ms_per_rank"]) == 4
    sw = out["strong_whole_matrix"]
    assert sw["columns"] == 2000 and sw["seconds"] > 0 and sw["kernel"] == "tile32", sw
    assert sw["item_space"]["kernel"] == "gram" and sw["item_space"]["seconds"] > 0, sw
    # --shard-gram: G was formed once by the four ranks together (row blocks, one broadcast per
    # block), so the solve itself built nothing
    assert len(sw["item_space"]["G_sharded_s"]) == 3 and sw["item_space"]["G_build_s"] == 0, sw


@pytest.mark.timeout(600, method="thread")
def test_bench_dry_run_of_an_eight_rank_step():
    """what eight ranks would each get in a step, measured on one device with
    no collective (bench.py --dry-run-world 8): the eight interleaved shards of a step's range
    solved one after the other -- kernel times within a few percent of each other, and the
    projected length of the driver's command reported against its limit.  (Scaled-down matrix
    here; the full-size figure is profiles/r04/dry_run_world8.json.)"""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-world", "8", "--workload", "c4",
           "--scale", "0.05", "--batch", "512", "--cpu-seconds", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=550, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["dry_run_world"] == 8 and len(out["ranks"]) == 8
    assert all(x["columns"] == 512 and x["kernel_ms"] > 0 for x in out["ranks"])
    # (16 tiles of a 50 000 x 5 000 matrix per shard run for milliseconds: their spread says
    # nothing; at full size it is 1.3 %, profiles/r04/dry_run_world8.json)
    assert out["kernel_ms_spread"] >= 0.0 and out["kernel_ms_mean"] > 0
    assert out["projected_command_s"]["total"] > 0
    item = out["item_space_whole_matrix"]
    assert item["G_row_block_build_ms"] > 0 and item["projected_step_s_with_sharded_G"] > 0, item
