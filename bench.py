#!/usr/bin/env python3
"""bench.py -- item-columns solved per second by the MI355X SLIM engine.

A "step" is one pass of the hot path (SLIMGPU_Learn: EstimateModelCD + SaveModel,
/root/reference/src/libslim/estimate.c:328-593) over one batch of B item columns of a
synthetic rating matrix that is already resident in HBM.  Default workload: BASELINE.json
configs[3], 1M users x 100K items, ~1e9 nnz, l1 = l2 = 1, optTol 1e-7 (the configuration
the metric is quoted on; it fits one GPU).  With N GPUs (one process per GPU) every rank
solves shard `rank` of N of the step's columns (granules of 32 columns of the cost-ordered
work list dealt round-robin; RCCL only for the one-off broadcast of R and the gather of the
learned columns).  Default: B = 8192 columns per GPU and step => weak scaling (at N = 8 a
step covers 65 536 of the 100 000 columns).  --scaling strong fixes the columns per step
(--batch 0: the whole matrix, north_star's target; 555 s per step on one GPU, which is why
it is not the default under the driver's 25-step run).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      algorithmic bytes (8G + 12D + 4U + 8 nnzW per column, from
                the engine's per-column counters) / solver-kernel time measured with HIP
                events on the engine's stream, against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a port of the reference's OpenMP CD path; the reference
                itself cannot be built here) timed on a bounded sample of the same columns
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s HBM3E (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SLIM_BENCH_WORKLOAD", "c4"),
                    help="c4 | c4-0.1pct | c5 | ml100k")
    ap.add_argument("--scale", type=float, default=float(os.environ.get("SLIM_BENCH_SCALE", "1")),
                    help="shrink both matrix dimensions (density kept); 1 = the named config")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("SLIM_BENCH_BATCH", "0")),
                    help="item columns per step and per GPU (0 = workload default)")
    ap.add_argument("--scaling", default=os.environ.get("SLIM_BENCH_SCALING", "weak"),
                    choices=["weak", "strong"],
                    help="weak: --batch columns per GPU and step; strong: --batch columns per "
                         "step in total, split over the GPUs (--batch 0 = the whole matrix)")
    ap.add_argument("--warmup-batch", type=int, default=0,
                    help="columns per GPU of an (untimed) warm-up step (0 = batch / 32)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ratings", action="store_true", help="ratings 1..5 instead of binary values")
    ap.add_argument("--kernel", type=int, default=0, help="slimgpu_kernel_et (0 = auto)")
    ap.add_argument("--cluster", type=int, default=int(os.environ.get("SLIM_BENCH_CLUSTER", "0")),
                    help="tile kernels: workgroups per tile, 1/2/4/8 (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0,
                    help="CPU-baseline budget of each one-thread mode (0 disables the whole leg)")
    ap.add_argument("--cpu-columns", type=int, default=int(os.environ.get("SLIM_BENCH_CPU_COLUMNS", "256")),
                    help="columns of the last step timed by the parallel CPU-baseline mode (rounds of "
                         "--cpu-threads columns each)")
    ap.add_argument("--cpu-threads", type=int, default=int(os.environ.get("SLIM_BENCH_CPU_THREADS", "32")),
                    help="threads of the parallel CPU-baseline modes (0 = all physical cores; on "
                         "the 128-core boxes of this pool one round then takes ~3 minutes)")
    ap.add_argument("--no-whole-matrix", action="store_true",
                    help="N >= 4: skip the extra whole-matrix (strong-scaling) step")
    ap.add_argument("--replicate", default="broadcast", choices=["broadcast", "generate"])
    ap.add_argument("--backend", default=os.environ.get("SLIM_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo lets several "
                         "ranks share one GPU in tests)")
    return ap.parse_args()


# Columns per GPU and step: 8192 = 256 tiles of 32.  (Round 2 used 8960, the size that fills the
# engine's rounds of 64 clusters best; a caller cannot choose that, so the default is a round
# number again.  Measured side by side, profiles/r03/c4_batch.txt: 8192 columns take 4.59 median
# tile times, 8960 take 4.90 -- 2.5 % apart per column; the rest of what separates two runs is
# the box, +-8 % from run to run.)  The whole-matrix step (--scaling strong --batch 0: 49 rounds,
# busy 0.97) is what either approximates inside the driver's 25 x step time box.
DEFAULT_BATCH = 8192

KERNEL_NAMES = {0: "auto", 1: "wave-lds", 2: "wave-hbm", 3: "tile32", 4: "tile16", 5: "gram"}


def main():
    args = parse_args()
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the one-rank-per-GPU job ourselves -- the
        # same command line under torch.distributed.run (what the driver does for N > 1)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: one rank per GPU" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the SLIM CD path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit("%d ranks but %d GPU(s): RCCL needs one device per rank" % (world, ndev))
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    # collectives carry device tensors over RCCL, host tensors over gloo
    cdev = dev if (world == 1 or args.backend == "nccl") else torch.device("cpu")

    from slim_amd import synth
    from slim_amd.distributed import broadcast_csr, gather_model
    from slim_amd.engine import DeviceMatrix

    # ---- the workload, resident in HBM before anything is timed ---------------------
    t_gen = time.time()
    if args.workload == "ml100k":
        from slim_amd.io import read_csr_text
        R = read_csr_text(os.path.join(ROOT, "tests", "golden", "ml100k-train.csr"))
        rowptr = torch.from_numpy(R.indptr.astype(np.int64)).to(dev)
        rowind = torch.from_numpy(R.indices.astype(np.int32)).to(dev)
        rowval = torch.from_numpy(R.data.astype(np.float32)).to(dev)
        nrows, ncols = R.shape
        name = "ml100k-train.csr 934x1683"
    else:
        nrows, ncols, target = synth.scaled(args.workload, args.scale) if args.scale != 1 \
            else synth.CONFIGS[args.workload]
        if rank == 0 or args.replicate == "generate":
            rowptr, rowind, rowval = synth.generate_csr(nrows, ncols, target, seed=args.seed,
                                                        ratings=args.ratings, device=dev)
            if not args.ratings:
                rowval = None  # implicit feedback: the C API's rowval == NULL path (4-byte nnz)
        else:
            rowptr = rowind = rowval = None
        if world > 1 and args.replicate == "broadcast":
            # the one data-path collective before the solve: R from rank 0 to every GPU
            if rank == 0 and cdev.type == "cpu":
                rowptr, rowind = rowptr.cpu(), rowind.cpu()
                rowval = rowval.cpu() if rowval is not None else None
            rowptr, rowind, rowval = broadcast_csr(rowptr, rowind, rowval, src=0)
            rowptr, rowind = rowptr.to(dev), rowind.to(dev)
            rowval = rowval.to(dev) if rowval is not None else None
        name = "synthetic %dx%d" % (nrows, ncols)
    nnz = int(rowind.numel())
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    t_stage = time.time()
    mat = DeviceMatrix.from_device_ptrs(nrows, ncols, rowptr.data_ptr(), rowind.data_ptr(),
                                        rowval.data_ptr() if rowval is not None else 0,
                                        keepalive=(rowptr, rowind, rowval), device=dev.index)
    t_stage = time.time() - t_stage
    ncols = mat.ncols

    # One step = one SLIMGPU_Learn per rank over shard `rank` of `world` of a range of item
    # columns (granules of 32 columns of the range's cost-ordered work list dealt round-robin:
    # every GPU gets the same mix of popular and unpopular items, no data-path collective).
    #   weak   (default): the range holds world x batch columns -> per-GPU work fixed
    #   strong          : the range holds batch columns in total (0 = the whole matrix)
    strong = args.scaling == "strong"
    per_gpu = args.batch or (ncols if args.workload == "ml100k" else DEFAULT_BATCH)
    if strong:
        span = min(ncols, args.batch or ncols)
    else:
        span = min(ncols, per_gpu * world)
    warm_span = min(span, world * (args.warmup_batch or max(256, per_gpu // 32)))
    opts = dict(l1r=1.0, l2r=1.0, optTol=1e-7, niters=10000, seed=args.seed, kernel=args.kernel)
    if args.cluster:
        opts["cluster"] = args.cluster

    def step(i, width):
        """Solve range i; with N > 1 also gather the learned columns on rank 0."""
        b = (i * span) % max(1, ncols - width + 1)
        W, st = mat.learn(col_begin=b, col_end=b + width, shard=(rank, world), **opts)
        if world > 1:
            W = gather_model(W, dst=0)  # the learned columns end up on rank 0
        return W, st, b

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):  # untimed: same code path on a smaller range
        step(i, warm_span)
    fence()
    t0 = time.perf_counter()
    acc = dict(kernel_ms=0.0, alg_bytes=0.0, G=0, D=0, U=0, nnzW=0, sweeps=0, gather_ms=0.0)
    last_b = None
    for i in range(args.steps):
        W, st, b = step(args.warmup + i, span)
        last_b = b  # W holds the columns of THIS step: the CPU leg must sample from it
        for k in acc:
            acc[k] += st[k]
    fence()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    # per-rank solver-kernel time of the timed steps (how even the shards were)
    rank_kernel_ms = [acc["kernel_ms"]]
    if world > 1:
        kt = torch.tensor([acc["kernel_ms"]], dtype=torch.float64, device=cdev)
        all_kt = [torch.zeros_like(kt) for _ in range(world)]
        dist.all_gather(all_kt, kt)
        rank_kernel_ms = [float(t.item()) for t in all_kt]

    # N >= 4: north_star's target itself as a secondary figure -- ONE step over the whole
    # matrix, its columns split over the GPUs (strong scaling), outside the timed region
    strong_whole = None
    if world >= 4 and not strong and args.workload != "ml100k" and not args.no_whole_matrix:
        fence()
        tw = time.perf_counter()
        Ww, stw = mat.learn(col_begin=0, col_end=ncols, shard=(rank, world), **opts)
        if world > 1:
            Ww = gather_model(Ww, dst=0)
        fence()
        tw = torch.tensor([time.perf_counter() - tw], dtype=torch.float64, device=cdev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        strong_whole = {"columns": int(ncols), "seconds": float(tw.item()),
                        "value": ncols / float(tw.item()), "unit": "item-columns/s",
                        "note": "one untimed-by-contract step: all item columns of the matrix, "
                                "sharded over the %d GPUs (strong scaling)" % world}
        del Ww

    if rank == 0:
        cols_total = args.steps * span
        kname = KERNEL_NAMES.get(st["kernel"], st["kernel"])
        achieved = acc["alg_bytes"] / (acc["kernel_ms"] * 1e-3) / 1e9 if acc["kernel_ms"] > 0 else 0.0
        kernel_s = acc["kernel_ms"] * 1e-3 / max(1, args.steps)
        traffic = os.environ.get("SLIM_BENCH_TRAFFIC_BYTES") or pmc_traffic(
            args, span if strong else per_gpu, kname, rowval is None, world)
        out = {
            "metric": "item-columns solved/sec (whole node)",
            "value": cols_total / elapsed,
            "unit": "item-columns/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "ranks": {"world": world, "backend": ("nccl (RCCL)" if args.backend == "nccl" else args.backend)
                      if world > 1 else "none (one process)",
                      "kernel_ms_per_rank": [round(v, 1) for v in rank_kernel_ms]},
            "vs_baseline": None,
            "dtype": "f32",
            "data": "fixture tests/golden/ml100k-train.csr" if args.workload == "ml100k" else "synthetic",
            "config": {
                "workload": "%s (%s), nnz %d, %s values, CD l1r=1 l2r=1 optTol=1e-7 "
                            "niters=10000; %d item columns per step %s"
                            % (args.workload, name, nnz,
                               "stored (all 1.0)" if args.workload == "ml100k" else
                               "ratings 1-5" if rowval is not None else "binary",
                               span if strong else per_gpu,
                               "in total" if strong else "per GPU"),
                "scale": args.scale, "seed": args.seed,
                "columns_per_step_per_gpu": span // world if strong else per_gpu,
                "columns_per_step": span,
                "warmup_columns_per_step": warm_span,
                "parallelism": "shards of the cost-ordered work list (32-column granules, "
                               "round-robin) over %d GPU(s), R replicated" % world,
                "kernel": kname,
                "generate_s": round(t_gen, 2), "stage_s": round(t_stage, 2),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": float(traffic) if traffic else None,
                # what the chip physically moved (PMC bytes) over the same kernel time
                "achieved_physical": float(traffic) / kernel_s / 1e9 if traffic and kernel_s > 0 else None,
                "frac_physical": float(traffic) / kernel_s / 1e9 / HBM_PEAK_GBS
                                 if traffic and kernel_s > 0 else None,
                "kernel_ms_per_launch": acc["kernel_ms"] / max(1, args.steps),
                "alg_bytes_per_launch": acc["alg_bytes"] / max(1, args.steps),
                "alg_bytes_per_column": acc["alg_bytes"] * world / max(1, args.steps * span),
                "G": acc["G"], "D": acc["D"], "U": acc["U"], "nnzW": acc["nnzW"],
                "sweeps": acc["sweeps"], "kernel_hash": kernel_hash(),
                "note": "rank 0's launches; algorithmic bytes = 8G+12D+4U+8nnzW per column "
                        "(binary: 4G+8D+4U+8nnzW), SURVEY.md 8(d)",
            },
        }
        if strong_whole is not None:
            out["strong_whole_matrix"] = strong_whole
        if world == 1:
            out["parity"] = ml100k_parity(dev.index)
        if world == 1 and args.cpu_seconds > 0 and args.workload != "ml100k":
            out["cpu_baseline"] = cpu_baseline(args, mat, rowptr, rowind, rowval, nrows, ncols,
                                               last_b, span, opts, W)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def ml100k_parity(device):
    """The second half of BASELINE.json's metric: HR@10 / ARHR on ml100k against the reference's
    0.3191 / 0.1504 (slim_predict.c:181-236 semantics).  Product path only:
    engine solve (cd, l1 = l2 = 1), GPU top-N scorer, hit counting in numpy."""
    import ctypes as C
    import numpy as np
    from slim_amd import _lib
    from slim_amd.engine import DeviceMatrix
    from slim_amd.io import read_csr_text
    lib = _lib.load()
    R = read_csr_text(os.path.join(ROOT, "tests", "golden", "ml100k-train.csr"))
    T = read_csr_text(os.path.join(ROOT, "tests", "golden", "ml100k-test.csr"))
    m = DeviceMatrix.from_scipy(R, device=device)
    t0 = time.perf_counter()
    h, st = m.learn(l1r=1.0, l2r=1.0, optTol=1e-7, niters=10000, seed=1, return_handle=True)
    t_learn = time.perf_counter() - t0
    hr_ = C.c_void_p()
    val = np.ascontiguousarray(R.data, np.float32)
    lib.Py_csr_wrapper(R.shape[0], np.ascontiguousarray(R.indptr, np.intp),
                       np.ascontiguousarray(R.indices, np.int32), val.ctypes.data_as(C.c_void_p),
                       C.byref(hr_))
    n = 10
    ids = np.full(R.shape[0] * n, -1, np.int32)
    sc = np.zeros(R.shape[0] * n, np.float32)
    rc = lib.SLIMGPU_Predict(n, h, hr_, ids, sc)
    ids = ids.reshape(-1, n)
    hits = arhr = 0.0
    nvalid = 0
    for u in range(T.shape[0]):
        test = T.indices[T.indptr[u]:T.indptr[u + 1]]
        if test.size == 0:
            continue
        nvalid += 1
        got = [r for r in range(n) if ids[u, r] >= 0 and ids[u, r] in test]
        hits += len(got) / float(test.size)                       # pyapi.c:309-366
        arhr += sum(1.0 / (1 + r) for r in got) / sum(1.0 / (1 + z) for z in range(test.size))
    lib.Py_csr_free(hr_)
    hh = C.c_void_p(h)
    lib.SLIM_FreeModel(C.byref(hh))
    m.close()
    hr10, arhr = hits / max(nvalid, 1), arhr / max(nvalid, 1)
    return {"dataset": "ml100k (tests/golden), cd l1r=1 l2r=1", "hr10": round(hr10, 4), "arhr": round(arhr, 4),
            "reference_hr10": 0.3191, "reference_arhr": 0.1504, "users": nvalid,
            "match": bool(rc == 1 and "%.4f" % hr10 == "0.3191" and "%.4f" % arhr == "0.1504"),
            "learn_ms": round(1e3 * t_learn, 2), "W_nnz": int(st["nnzW"])}


KERNEL_SOURCES = ("cd_tile.hpp", "cd_wave.hpp", "cd_perm.hpp", "engine.hip", "tile_inst.hpp")


def kernel_hash():
    """Fingerprint of the solver sources: a PMC figure collected for another build of the
    kernels must not be reported for this one."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "slim_amd", "csrc", name)) as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # comments and layout do not
        text = re.sub(r"//[^\n]*", "", text)                       # change the machine code
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def pmc_traffic(args, columns, kernel, binary, world=1):
    """HBM bytes per launch from the PMC counters.  They cannot be collected inside this
    process (rocprofv3 wraps the whole command, one --pmc pass per counter), so the figure
    measured for this exact configuration AND this build of the kernels
    (scripts/collect_profiles.sh -> profiles/pmc_traffic.json) is looked up; anything else --
    another configuration, or sources edited since the collection -- reports null."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            entries = json.load(f)["entries"]
    except (OSError, ValueError, KeyError):
        return None
    if world != 1:
        return None
    for e in entries:
        m = e["match"]
        if (m["workload"] == args.workload and float(m["scale"]) == float(args.scale) and
                m["columns_per_step_per_gpu"] == columns and m["kernel"] == kernel and
                bool(m["binary"]) == bool(binary) and int(m.get("seed", 1)) == int(args.seed) and
                e.get("kernel_hash") == kernel_hash()):
            return e["traffic_bytes_per_launch"]
    return None


def rowind_col_nnz(R, col):
    """Number of ratings of one item (column of the CSR matrix R)."""
    import numpy as np
    return int(np.count_nonzero(R.indices == col))


def cpu_baseline(args, mat, rowptr, rowind, rowval, nrows, ncols, b, span, opts, W_gpu):
    """The CPU restatement of the reference's OpenMP CD path (oracle/slim_oracle.c:
    estimate.c:328-558 + cd.c, reference arithmetic: fp64, three passes per visit) timed on
    this box's host cores on a seeded sample of the columns the GPU just solved -- SURVEY.md
    8(d): faithful mode (libc rand() shuffle, full-scan aTy, estimate.c:412-421 / cd.c:76-86)
    and thread-local-PRNG + Gram-column-aTy mode, on all physical cores (one column per core)
    and on one thread; estimate phase only (the reference's LearnTmr), setup excluded.
    Checker use: the same leg verifies the GPU's columns -- one whole tile of the step against
    the oracle walking that tile in the kernel's visiting order, and the timing sample
    against the oracle's own order (order-to-order envelope)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import scipy.sparse as sp
    import slim_oracle as O

    binary = rowval is None
    vals = np.ones(rowind.numel(), np.float32) if binary else rowval.cpu().numpy()
    R = sp.csr_matrix((vals, rowind.cpu().numpy(), rowptr.cpu().numpy()), shape=(nrows, ncols))
    cores, threads = O.physical_cores()
    O.cache_setup(True)   # one transpose of R for all the oracle calls below (setup is untimed)
    kw = dict(l1r=opts["l1r"], l2r=opts["l2r"], optTol=opts["optTol"], maxniters=opts["niters"],
              binary=binary, chunk=1)
    gram = dict(order=O.ORDER_LOCAL, seed=opts["seed"], aty=O.ATY_GRAM)
    faithful = dict(order=O.ORDER_GLIBC, aty=O.ATY_FULLSCAN, srand=1)
    rng = np.random.default_rng(args.seed)
    pool = b + rng.permutation(span)

    def timed(cols, nthreads, mode):
        Wc = O.learn_cd(R, cols=np.sort(cols).astype(np.int32), nthreads=nthreads, **kw, **mode)
        return Wc, O.learn_seconds()

    # one column on one thread sizes everything else
    _, t1 = timed(pool[:1], 1, gram)
    n1 = int(max(1, min(8, args.cpu_seconds // max(t1, 1e-3))))
    res = {}
    _, t = timed(pool[:n1], 1, gram)
    res["gram_localprng_1thread"] = {"value": n1 / t, "columns": n1, "seconds": round(t, 3), "threads": 1}
    _, t = timed(pool[:n1], 1, faithful)
    res["fullscan_rand_1thread"] = {"value": n1 / t, "columns": n1, "seconds": round(t, 3), "threads": 1}
    # the parallel modes: one column per thread and round, on --cpu-threads threads (default 32:
    # on the 2 x 64-core hosts of this pool 128 concurrent columns thrash the memory system --
    # 0.73 col/s on 128 threads against ~2.5 on 32, profiles/r02/cpu_baseline_128threads.txt --
    # and one such round takes three minutes); a round takes longer than one column alone, so
    # the round count comes from a first round
    # SURVEY.md 8(d) asks for a sample of >= 512 columns; --cpu-columns (default 256, ~100 s on
    # 32 threads) are timed round by round so that the spread is visible: `value` is all
    # columns over all seconds, `rounds` lists every round's rate
    use = max(1, min(args.cpu_threads or cores, cores, span))
    rounds = int(max(1, min(span // use, -(-max(args.cpu_columns, use) // use))))
    sample = pool[:use * rounds]
    parts, t = [], 0.0
    round_rates = []
    for k in range(rounds):
        Wk, tk = timed(sample[k * use:(k + 1) * use], use, gram)
        parts.append(Wk)
        t += tk
        round_rates.append(round(use / tk, 3))
    Wc = parts[0]
    for Wk in parts[1:]:
        Wc = Wc + Wk   # disjoint columns
    srt = sorted(round_rates)
    res["gram_localprng_allcores"] = {"value": sample.size / t, "columns": int(sample.size),
                                      "seconds": round(t, 3), "threads": use,
                                      "rounds": round_rates, "round_min": srt[0],
                                      "round_median": srt[len(srt) // 2], "round_max": srt[-1]}
    _, tf = timed(pool[:use], use, faithful)
    res["fullscan_rand_allcores"] = {"value": use / tf, "columns": use, "seconds": round(tf, 3),
                                     "threads": use}
    # parity of the GPU's columns
    sample = np.sort(sample)
    Wg = sp.csc_matrix(W_gpu)
    diff = abs(Wg[:, sample] - Wc[:, sample])
    d_sample = float(diff.max()) if diff.nnz else 0.0
    w_max = float(abs(Wc[:, sample]).max()) if Wc[:, sample].nnz else 0.0
    worst = {}
    if diff.nnz:   # where the largest difference sits, and how large that column's coefficients are
        dc = diff.tocoo()
        k = int(dc.data.argmax())
        col = int(sample[dc.col[k]])
        worst = {"column": col, "row": int(dc.row[k]),
                 "column_max_abs_W": float(abs(Wc[:, [col]]).max()),
                 "column_nnz": int(rowind_col_nnz(R, col))}
    # two valid visiting orders stop at slightly different points at optTol 1e-7 (the reference
    # differs from itself by 0.25 % of max|W| across shuffle seeds on ml100k; 1.4e-5 of 7.5e-3 on a
    # C4 median tile, profiles/r02/fullsize_parity.txt; 1.9e-4 over 256 C4 columns that include
    # the most popular items): the sample check allows 2 % of the sample's largest coefficient
    tol_sample = max(1e-4, 0.02 * w_max)
    d_tile = None
    cost = mat.column_cost()
    cols = np.arange(b, b + span)
    order = cols[np.argsort(-cost[cols], kind="stable")].astype(np.int32)
    ntiles = (order.size + 31) // 32
    g = ntiles // 2
    tile = order[g * 32:(g + 1) * 32]
    if args.kernel in (0, 3) and order.size >= 64:
        Wt = O.learn_cd_tile(R, tileP=32, order=order, maxniters=opts["niters"], seed=opts["seed"],
                             nthreads=min(32, threads), binary=binary, tiles=(g, 1),
                             l1r=opts["l1r"], l2r=opts["l2r"], optTol=opts["optTol"])
        dt = abs(Wg[:, tile] - Wt[:, tile])
        d_tile = float(dt.max()) if dt.nnz else 0.0
    O.cache_setup(False)
    best = res["gram_localprng_allcores"]
    return {
        "value": best["value"], "unit": "item-columns/s", "cores": best["threads"], "kind": "port",
        "host": "%s, %d physical cores / %d hardware threads" % (O.cpu_model(), cores, threads),
        "sample": "%d of the %d columns of the last GPU step (seeded choice), %.1f s of CPU "
                  "work, estimate phase only; oracle/slim_oracle.c, reference arithmetic (fp64, "
                  "3-pass), thread-local PRNG shuffle + Gram-column aTy, OpenMP one column per "
                  "thread on %d threads" % (best["columns"], span, best["seconds"], best["threads"]),
        "modes": res,
        "parity": {
            "tile_order_max_abs_dW": d_tile, "tile": "tile %d of %d of the last step" % (g, ntiles),
            "tile_order_tolerance": 2e-5,
            "sample_max_abs_dW": d_sample, "sample_max_abs_W": w_max, "sample_worst": worst,
            "sample_tolerance": tol_sample,
            "parity_ok": bool((d_tile is None or d_tile <= 2e-5) and d_sample <= tol_sample),
            "note": "tile: GPU vs oracle_learn_cd_tile walking the same tile in the kernel's "
                    "visiting order (visit-for-visit); sample: GPU (tile order) vs the oracle's "
                    "own per-item order at optTol 1e-7 -- order-to-order envelope (2 % of the "
                    "sample's largest coefficient allowed; 1.4e-5 of 7.5e-3 on a C4 median tile, "
                    "profiles/r02/fullsize_parity.txt)",
        },
    }


if __name__ == "__main__":
    main()
