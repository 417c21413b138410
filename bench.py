#!/usr/bin/env python3
"""bench.py -- item-columns solved per second by the MI355X SLIM engine.

A "step" is one pass of the hot path (SLIMGPU_Learn: EstimateModelCD + SaveModel,
/root/reference/src/libslim/estimate.c:328-593) over one batch of B item columns of a
synthetic rating matrix that is already resident in HBM.  Default workload: BASELINE.json
configs[3], 1M users x 100K items, ~1e9 nnz, l1 = l2 = 1, optTol 1e-7 (the configuration
the metric is quoted on; it fits one GPU).  Item columns are block-partitioned over the
ranks (one process per GPU, RCCL only for the one-off broadcast of R and the gather of
the learned columns); per-GPU work is fixed as N grows => weak scaling.

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
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ratings", action="store_true", help="ratings 1..5 instead of binary values")
    ap.add_argument("--kernel", type=int, default=0, help="slimgpu_kernel_et (0 = auto)")
    ap.add_argument("--cluster", type=int, default=int(os.environ.get("SLIM_BENCH_CLUSTER", "0")),
                    help="tile kernels: workgroups per tile, 1/2/4/8 (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="CPU-baseline budget (0 disables the leg)")
    ap.add_argument("--replicate", default="broadcast", choices=["broadcast", "generate"])
    return ap.parse_args()


def main():
    args = parse_args()
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the SLIM CD path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from slim_amd import synth
    from slim_amd.distributed import broadcast_csr, gather_model, partition_columns
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
            rowptr, rowind, rowval = broadcast_csr(rowptr, rowind, rowval, src=0)
        name = "synthetic %dx%d" % (nrows, ncols)
    nnz = int(rowind.numel())
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    t_stage = time.time()
    mat = DeviceMatrix.from_device_ptrs(nrows, ncols, rowptr.data_ptr(), rowind.data_ptr(),
                                        rowval.data_ptr() if rowval is not None else 0,
                                        keepalive=(rowptr, rowind, rowval), device=local_rank)
    t_stage = time.time() - t_stage
    ncols = mat.ncols

    blocks = partition_columns(mat.column_cost(), world)
    cb, ce = blocks[rank]
    batch = args.batch or (ncols if args.workload == "ml100k" else 8192)
    batch = max(1, min(batch, ce - cb))
    opts = dict(l1r=1.0, l2r=1.0, optTol=1e-7, niters=10000, seed=args.seed, kernel=args.kernel)
    if args.cluster:
        opts["cluster"] = args.cluster

    def step(i):
        """Solve batch i of this rank's block; with N > 1 also gather the learned columns."""
        b = cb + (i * batch) % max(1, (ce - cb) - batch + 1)
        W, st = mat.learn(col_begin=b, col_end=b + batch, **opts)
        if world > 1:
            W = gather_model(W, dst=0)  # the learned columns end up on rank 0
        return W, st, b

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    acc = dict(kernel_ms=0.0, alg_bytes=0.0, G=0, D=0, U=0, nnzW=0, sweeps=0, gather_ms=0.0)
    last_b = None
    for i in range(args.steps):
        W, st, b = step(args.warmup + i)
        last_b = b  # W holds the columns of THIS step: the CPU leg must sample from it
        for k in acc:
            acc[k] += st[k]
    fence()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0:
        cols_total = world * args.steps * batch
        achieved = acc["alg_bytes"] / (acc["kernel_ms"] * 1e-3) / 1e9 if acc["kernel_ms"] > 0 else 0.0
        traffic = os.environ.get("SLIM_BENCH_TRAFFIC_BYTES") or pmc_traffic(
            args, batch, {0: "auto", 1: "wave-lds", 2: "wave-hbm", 3: "tile32", 4: "tile16"}.get(
                st["kernel"]), rowval is None)
        out = {
            "metric": "item-columns solved/sec (whole node)",
            "value": cols_total / elapsed,
            "unit": "item-columns/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "fixture tests/golden/ml100k-train.csr" if args.workload == "ml100k" else "synthetic",
            "config": {
                "workload": "%s (%s), nnz %d, %s values, CD l1r=1 l2r=1 optTol=1e-7 "
                            "niters=10000; %d item columns per step per GPU"
                            % (args.workload, name, nnz,
                               "stored (all 1.0)" if args.workload == "ml100k" else
                               "ratings 1-5" if rowval is not None else "binary", batch),
                "scale": args.scale, "columns_per_step_per_gpu": batch,
                "parallelism": "columns block-partitioned over %d GPU(s), R replicated" % world,
                "kernel": {0: "auto", 1: "wave-lds", 2: "wave-hbm", 3: "tile32", 4: "tile16"}.get(st["kernel"], st["kernel"]),
                "generate_s": round(t_gen, 2), "stage_s": round(t_stage, 2),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": float(traffic) if traffic else None,
                "kernel_ms_per_launch": acc["kernel_ms"] / max(1, args.steps),
                "alg_bytes_per_launch": acc["alg_bytes"] / max(1, args.steps),
                "alg_bytes_per_column": acc["alg_bytes"] / max(1, args.steps * batch),
                "G": acc["G"], "D": acc["D"], "U": acc["U"], "nnzW": acc["nnzW"],
                "sweeps": acc["sweeps"],
            },
        }
        if world == 1:
            out["parity"] = ml100k_parity(local_rank)
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, rowptr, rowind, rowval, nrows, ncols,
                                               last_b, batch, opts, W)
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


def pmc_traffic(args, batch, kernel, binary):
    """HBM bytes per launch from the PMC counters.  They cannot be collected inside this
    process (rocprofv3 wraps the whole command, one --pmc pass per counter), so the figure
    measured for this exact configuration is read from profiles/pmc_traffic.json; any other
    configuration reports null."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            entries = json.load(f)["entries"]
    except (OSError, ValueError, KeyError):
        return None
    for e in entries:
        m = e["match"]
        if (m["workload"] == args.workload and float(m["scale"]) == float(args.scale) and
                m["columns_per_step_per_gpu"] == batch and m["kernel"] == kernel and
                bool(m["binary"]) == bool(binary)):
            return e["traffic_bytes_per_launch"]
    return None


def cpu_baseline(args, rowptr, rowind, rowval, nrows, ncols, b, batch, opts, W_gpu):
    """Time the CPU oracle (port of estimate.c:328-558 + cd.c) on a bounded sample of the
    columns the GPU just solved, on this box's host cores.  Checker use only: its W is
    compared with the GPU's for the sampled columns."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import scipy.sparse as sp
    import slim_oracle as O

    vals = np.ones(rowind.numel(), np.float32) if rowval is None else rowval.cpu().numpy()
    R = sp.csr_matrix((vals, rowind.cpu().numpy(), rowptr.cpu().numpy()), shape=(nrows, ncols))
    threads = O.max_threads()
    # fp32=True: the oracle's fused one-pass arithmetic (the same operation count as the GPU
    # kernels; ~2-3x faster on the CPU than the reference's 3-pass fp64 form, i.e. the
    # stronger baseline)
    kw = dict(l1r=opts["l1r"], l2r=opts["l2r"], optTol=opts["optTol"], maxniters=opts["niters"],
              order=O.ORDER_PERM, seed=opts["seed"], aty=O.ATY_GRAM, binary=rowval is None,
              fp32=True)
    rng = np.random.default_rng(args.seed)
    pool = b + rng.permutation(batch)
    # probe: 4 columns on 4 threads, then size the sample (one column per thread) so that
    # the timed run stays near the budget; on a bandwidth-bound host the time of a round
    # grows with the number of threads streaming R at once
    probe = np.sort(pool[:min(4, batch)]).astype(np.int32)
    t0 = time.perf_counter()
    O.learn_cd(R, cols=probe, nthreads=len(probe), **kw)
    t_probe = time.perf_counter() - t0
    use = threads
    while use > 8 and t_probe * (1.0 + use / 24.0) > args.cpu_seconds:
        use //= 2
    use = max(1, min(use, batch))
    rounds = int(max(1, min(batch // use, args.cpu_seconds // max(t_probe * (1.0 + use / 24.0), 1e-3))))
    sample = np.sort(pool[:min(batch, use * rounds)]).astype(np.int32)
    threads = use
    t0 = time.perf_counter()
    Wc = O.learn_cd(R, cols=sample, nthreads=use, **kw)
    t_cpu = time.perf_counter() - t0
    diff = abs(sp.csc_matrix(W_gpu)[:, sample] - Wc[:, sample])
    return {
        "value": sample.size / t_cpu, "unit": "item-columns/s", "cores": threads, "kind": "port",
        "sample": "%d of the %d columns of the last GPU step (seeded choice), %.1f s of CPU "
                  "work; oracle/slim_oracle.c (fused fp32 arithmetic) with OpenMP over columns on "
                  "%d of the host's %d hardware threads, Gram-column aTy"
                  % (sample.size, batch, t_cpu, use, O.max_threads()),
        "max_abs_dW_vs_gpu": float(diff.max()) if diff.nnz else 0.0,
    }


if __name__ == "__main__":
    main()
