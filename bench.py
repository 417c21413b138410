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
and, at N = 1 on the default workload, two secondary figures under their own keys (never
`value`; SURVEY.md 8(d)'s formula does not price them):
  item_space_step  up to five of the timed steps' column ranges once more, from scratch, in item
                   space (cd_gramr.hpp; G = R^T R built inside the first one's measured time and
                   reported apart and amortised), with the difference of the models
  item_space_grid  the first pairs of the C5 model-selection grid (BASELINE.json configs[4])
                   on the path the engine takes for a grid
  item_space_whole_matrix  every item column of the matrix once (north_star's target at N = 1)
The command budgets itself against the driver's 1800 s (SLIM_BENCH_WALL_BUDGET): the extras
and the CPU leg shrink or drop out, with a note, rather than overrun.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s HBM3E (spec)
T_START = time.time()
# the driver gives the whole command 1800 s; the CPU leg (last thing in the run) fits itself
# into what the timed steps left of this
WALL_BUDGET_S = float(os.environ.get("SLIM_BENCH_WALL_BUDGET", "1680"))


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
    ap.add_argument("--cpu-columns", type=int, default=int(os.environ.get("SLIM_BENCH_CPU_COLUMNS", "512")),
                    help="columns of the last step timed by the parallel CPU-baseline mode (rounds of "
                         "--cpu-threads columns each)")
    ap.add_argument("--cpu-threads", type=int, default=int(os.environ.get("SLIM_BENCH_CPU_THREADS", "32")),
                    help="threads of the parallel CPU-baseline modes (0 = all physical cores; on "
                         "the 128-core boxes of this pool one round then takes ~3 minutes)")
    ap.add_argument("--dry-run-world", type=int, default=0,
                    help="one device, no collectives: solve the N shards one rank-of-N each would get "
                         "in a step, one after the other, and report the per-rank kernel times, their "
                         "spread and the projected length of the driver's N-GPU command")
    ap.add_argument("--no-item-space", action="store_true",
                    help="N = 1: skip the secondary figure (first pairs of the C5 grid in item space)")
    ap.add_argument("--shard-gram", action="store_true",
                    help="N >= 4, the whole-matrix step in item space: form G = R^T R once by all ranks (row blocks + "
                         "one RCCL broadcast per block, slim_amd.distributed.build_gram_sharded) instead of once per rank")
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

    # The timed steps are from-scratch solves by the residual kernels (SURVEY.md 8(d) prices
    # THEIR traffic): nothing is carried from one step to the next -- neither the screen sums a
    # second solve of the same columns could reuse (steps over ml100k or the whole matrix repeat
    # their work list) nor the engine's switch to item-space CD for repeated solves.  --kernel 5
    # asks for item-space CD explicitly; it is reported under its own key otherwise.
    if args.kernel != 5:
        os.environ["SLIM_GPU_NO_GRAM"] = "1"
    # `value` and `roofline` price the RESIDUAL kernel (SURVEY.md 8(d)'s formula is its traffic):
    # it is pinned explicitly for the timed steps of the synthetic workloads, because the
    # engine's own choice (SLIMGPU_KERNEL_AUTO, what SLIM_Learn takes by default) is item space
    # there since round 5 -- that path is reported beside it as item_space_step / item_space_grid
    # with its own byte model.  --kernel 5 times item space itself; --kernel 0 on ml100k is the
    # one-wavefront-per-item kernel.
    timed_kernel = args.kernel
    if timed_kernel == 0 and args.workload != "ml100k":
        timed_kernel = 3

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
    opts = dict(l1r=1.0, l2r=1.0, optTol=1e-7, niters=10000, seed=args.seed, kernel=timed_kernel)
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

    if args.dry_run_world > 1 and world == 1:
        print(json.dumps(dry_run(args, mat, ncols, per_gpu, opts, nnz)))
        return

    for i in range(args.warmup):  # untimed: same code path on a smaller range
        step(i, warm_span)
    fence()
    t0 = time.perf_counter()
    acc = dict(kernel_ms=0.0, alg_bytes=0.0, G=0, D=0, U=0, nnzW=0, sweeps=0, gather_ms=0.0)
    last_b = None
    step_begins = []
    for i in range(args.steps):
        W, st, b = step(args.warmup + i, span)
        last_b = b  # W holds the columns of THIS step: the CPU leg must sample from it
        step_begins.append(b)
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
    # (only when the command's wall budget has room for it: its length is the timed step's times
    # the ratio of the columns per GPU, and the CPU-free N > 1 run ends right after it)
    whole_est = (elapsed / max(1, args.steps)) * (ncols / float(world)) / max(1, per_gpu) * 1.25
    go_whole = world >= 4 and not strong and args.workload != "ml100k" and not args.no_whole_matrix
    if go_whole:  # one decision for all ranks (their clocks started at different moments)
        room = torch.tensor([WALL_BUDGET_S - (time.time() - T_START) - whole_est - 60.0],
                            dtype=torch.float64, device=cdev)
        dist.all_reduce(room, op=dist.ReduceOp.MIN)
        go_whole = float(room.item()) > 0.0
    if go_whole:
        def whole(kernel, env, shard_gram=False):
            """One step over all item columns, sharded over the ranks; never raises (a failure in an
            extra step must not cost the line its timed figures)."""
            saved = {k: os.environ.get(k) for k in env}
            try:
                for k, v in env.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                fence()
                tw = time.perf_counter()
                err = None
                gram_s = None
                if shard_gram:   # (collectives inside: opt-in, --shard-gram)
                    from slim_amd.distributed import build_gram_sharded
                    gram_s = build_gram_sharded(mat)
                try:   # the solve is local to the rank: a failure here is agreed on before any collective
                    Ww, stw = mat.learn(col_begin=0, col_end=ncols, shard=(rank, world), **dict(opts, kernel=kernel))
                except Exception as e:   # noqa: BLE001
                    err = "%s: %s" % (type(e).__name__, e)
                okf = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=cdev)
                dist.all_reduce(okf, op=dist.ReduceOp.MIN)
                if float(okf.item()) < 1.0:
                    return {"error": err or "another rank failed"}
                if world > 1:
                    Ww = gather_model(Ww, dst=0)
                fence()
                tw = torch.tensor([time.perf_counter() - tw], dtype=torch.float64, device=cdev)
                dist.all_reduce(tw, op=dist.ReduceOp.MAX)
                del Ww
                return {"columns": int(ncols), "seconds": float(tw.item()),
                        "value": ncols / float(tw.item()), "unit": "item-columns/s",
                        "kernel": KERNEL_NAMES.get(stw["kernel"], stw["kernel"]),
                        "G_build_s": round(stw["gram_build_ms"] * 1e-3, 2),
                        "G_build_split_s": gram_split(stw) if stw["gram_build_ms"] else None,
                        "G_sharded_s": [round(x, 3) for x in gram_s] if gram_s else None}
            except Exception as e:   # noqa: BLE001
                return {"error": "%s: %s" % (type(e).__name__, e)}
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        strong_whole = whole(opts["kernel"], {})
        strong_whole["note"] = ("one untimed-by-contract step: all item columns of the matrix, sharded "
                                "over the %d GPUs (strong scaling), residual kernel" % world)
        # the same step in item space (every rank builds G = R^T R for itself, then solves its shard)
        if "seconds" in strong_whole:
            room = torch.tensor([WALL_BUDGET_S - (time.time() - T_START) - 0.6 * strong_whole["seconds"] - 90.0],
                                dtype=torch.float64, device=cdev)
            dist.all_reduce(room, op=dist.ReduceOp.MIN)
            if float(room.item()) > 0.0:
                strong_whole["item_space"] = whole(5, {"SLIM_GPU_NO_GRAMCD": None, "SLIM_GPU_NO_GRAM": None},
                                                   shard_gram=args.shard_gram)

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
                "carried_between_steps": "nothing (screen-sum cache and the automatic switch to "
                                         "item-space CD are off for the timed steps: kernel pinned)"
                                         if args.kernel != 5 else "G = R^T R, built by the first step",
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
        # the engine's default path on this workload, under its own keys (~60 s together; they run
        # BEFORE the CPU leg, which fits itself into whatever is left of the wall budget)
        if world == 1 and args.workload == "c4" and args.scale == 1 and args.kernel != 5 \
                and not args.no_item_space and WALL_BUDGET_S - (time.time() - T_START) > 200:
            out["item_space_step"] = item_space_step(args, mat, step_begins, span, opts, W)
            out["item_space_grid"] = item_space_grid(args, dev)
            # north_star's literal target at N = 1: every item column of the matrix, on the path
            # SLIM_Learn takes (G is on the handle by now: its cost is item_space_step's G_build_s)
            left = WALL_BUDGET_S - (time.time() - T_START)
            if left > 150 and not args.no_whole_matrix:
                out["item_space_whole_matrix"] = item_space_whole_matrix(args, mat, ncols, opts,
                                                                         out["item_space_step"])
            else:
                out["item_space_whole_matrix"] = {"skipped": "%.0f s of the wall budget left (needs 150)" % left}
        if world == 1 and args.cpu_seconds > 0 and args.workload != "ml100k":
            out["cpu_baseline"] = cpu_baseline(args, mat, rowptr, rowind, rowval, nrows, ncols,
                                               last_b, span, opts, W)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def dry_run(args, mat, ncols, per_gpu, opts, nnz):
    """What an N-GPU run of this command would hand to each rank, measured on ONE device without
    any collective: shard r of N of a step's range (N x batch columns, weak scaling), r = 0..N-1,
    solved one after the other.  Reports the solver-kernel time of every rank-to-be, their
    spread (how even the interleaved 32-column granules are) and the projected length of the
    driver's command (warm-up + timed steps at the slowest rank's pace, the extra whole-matrix
    step, the broadcast of R at one xGMI link's rate) against its 1800 s limit."""
    N = args.dry_run_world
    span = min(ncols, per_gpu * N)
    ranks = []
    # (one small untimed solve first: the column splits of the cluster geometry are built once)
    mat.learn(col_begin=0, col_end=max(256, span // 32), shard=(0, N), **opts)
    for r in range(N):
        t0 = time.perf_counter()
        _, st = mat.learn(col_begin=0, col_end=span, shard=(r, N), **opts)
        ranks.append({"rank": r, "columns": int(st["ncols_solved"]), "kernel_ms": round(st["kernel_ms"], 1),
                      "wall_ms": round(1e3 * (time.perf_counter() - t0), 1),
                      "alg_bytes": st["alg_bytes"]})
    # the same for the N >= 4 extra in item space (strong_whole_matrix.item_space): every rank builds
    # G = R^T R for itself (measured once here) and solves shard r of ALL the columns
    item = None
    if not args.no_item_space:
        saved = {k: os.environ.pop(k, None) for k in ("SLIM_GPU_NO_GRAM", "SLIM_GPU_NO_GRAMCD")}
        try:
            # (G formed once by the ranks together, --shard-gram: what rank 0's row block costs, and the commit)
            from slim_amd.distributed import gram_blocks
            import torch as _t
            gb, ge = gram_blocks(ncols, N)[0]
            _t.cuda.synchronize()
            t0 = time.perf_counter()
            mat.gram_build_rows(gb, ge)
            _t.cuda.synchronize()
            block_ms = 1e3 * (time.perf_counter() - t0)
            ir = []
            for r in range(N):
                t0 = time.perf_counter()
                _, st = mat.learn(col_begin=0, col_end=ncols, shard=(r, N), **dict(opts, kernel=5))
                ir.append({"rank": r, "columns": int(st["ncols_solved"]), "kernel_ms": round(st["kernel_ms"], 1),
                           "G_build_ms": round(st["gram_build_ms"], 1),
                           "wall_ms": round(1e3 * (time.perf_counter() - t0), 1)})
            g_ms = max(x["G_build_ms"] for x in ir)
            solve_ms = max(x["wall_ms"] - x["G_build_ms"] for x in ir)
            free_b, total_b = __import__("torch").cuda.mem_get_info()
            item = {"ranks": ir, "G_build_ms_per_rank": g_ms,
                    "projected_step_s": round((g_ms + solve_ms) * 1e-3, 1),
                    "G_row_block_build_ms": round(block_ms, 1),
                    "projected_step_s_with_sharded_G": round((block_ms + 100.0 + 330.0 + solve_ms) * 1e-3, 1),
                    "hbm_in_use_gb_with_G": round((total_b - free_b) / 1e9, 1), "hbm_total_gb": round(total_b / 1e9, 1),
                    "note": "whole matrix in item space at N ranks: each rank pays the G build (measured once, on "
                            "rank 0's shard) plus its shard; memory = R + G (floats + byte planes) + slabs on one GPU; "
                            "with_sharded_G (--shard-gram): rank 0's row block of G built alone + ~0.1 s for the N block "
                            "broadcasts (5 GB per link at N = 8) + 0.33 s of byte planes + the shard"}
        except Exception as e:   # noqa: BLE001
            item = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            for k, v in saved.items():
                if v is not None:
                    os.environ[k] = v
    km = [x["kernel_ms"] for x in ranks]
    wm = [x["wall_ms"] for x in ranks]
    mean = sum(km) / len(km)
    step_s = max(wm) * 1e-3
    steps, warmup = 20, 5                      # the driver's command line
    bytes_R = 8.0 * (mat.nrows + 1) + 4.0 * nnz
    bcast_s = bytes_R / 50e9                   # ring broadcast over xGMI: one link's ~50 GB/s effective
    whole_s = step_s * (ncols / float(N)) / max(1, per_gpu) * 1.25
    warm_s = warmup * step_s / 32.0 * 2.0      # warm-up steps run 1/32 of a step's range
    item_s = item["projected_step_s"] if item and "projected_step_s" in item else 0.0
    total = 60.0 + bcast_s + warm_s + steps * step_s + whole_s + item_s
    return {"dry_run_world": N, "columns_per_rank_and_step": per_gpu, "range": span,
            "ranks": ranks, "item_space_whole_matrix": item, "kernel_ms_mean": round(mean, 1),
            "kernel_ms_spread": round((max(km) - min(km)) / mean, 4),
            "projected_command_s": {"start_up_and_generate": 60.0, "broadcast_R": round(bcast_s, 2),
                                    "warmup": round(warm_s, 1), "timed_steps": round(steps * step_s, 1),
                                    "whole_matrix_step": round(whole_s, 1),
                                    "whole_matrix_step_item_space": round(item_s, 1), "total": round(total, 1),
                                    "limit": 1800.0, "fits": bool(total < 1800.0),
                                    "whole_matrix_step_runs": bool(total < WALL_BUDGET_S)},
            "note": "one device, ranks solved one after the other: no RCCL, no peer copies -- what "
                    "is measured is the evenness of the shards and the step time they imply"}


def gram_split(st):
    """The parts of SLIMGPU_LastStats.gram_build_ms, in seconds (slim_gpu.h): the allocation of G
    (a first 40 GB hipMalloc costs seconds on some boxes and nothing on others -- what made the same
    build 2.7 s in one record and 5.7 s in another), the sums (the whole nested solve, and its kernel
    alone), the byte planes."""
    return {"allocation": round(st.get("gram_alloc_ms", 0.0) * 1e-3, 2),
            "sums": round(st.get("gram_sums_ms", 0.0) * 1e-3, 2),
            "sums_kernel": round(st.get("gram_sums_kernel_ms", 0.0) * 1e-3, 2),
            "byte_planes": round(st.get("gram_pack_ms", 0.0) * 1e-3, 2)}


def item_space_step(args, mat, begins, span, opts, W_res, nmax=5):
    """Secondary figure, under its own key: up to `nmax` of the timed steps once more -- the same
    column ranges of the same matrix, from scratch -- on the path SLIM_Learn takes by default
    (SLIMGPU_KERNEL_AUTO: item space, cd_gramr.hpp).  The LAST timed step comes first: it builds
    G = R^T R of the whole matrix inside its measured time (nothing is carried in) and is compared
    with the timed step's model (the two kernels walk the same visiting order: fp32 rounding only);
    the other ranges then run with G on the handle, as every solve after the first does.  `value` /
    `seconds` are the first (from-scratch) step's, as in round 5; `steps` has every range, `mean` /
    `min` / `max` over their solve times, `value_amortised` = all columns / (all solves + G once).
    The roofline object uses the kernel's own byte model (bytes of G streamed,
    SLIMGPU_LastStats.gram_bytes) over its HIP-event time, summed over the ranges."""
    import scipy.sparse as sp
    saved = {k: os.environ.pop(k, None) for k in ("SLIM_GPU_NO_GRAM", "SLIM_GPU_NO_GRAMCD")}
    try:
        order = [begins[-1]] + [x for x in dict.fromkeys(reversed(begins[:-1]))][:nmax - 1]
        recs, first = [], None
        bytes_sum = ks_sum = 0.0
        for j, b in enumerate(order):
            t0 = time.perf_counter()
            Wi, st = mat.learn(col_begin=b, col_end=b + span, **dict(opts, kernel=0))
            dt = time.perf_counter() - t0
            ks = st["kernel_ms"] * 1e-3
            if j == 0:
                d = abs(sp.csc_matrix(Wi) - sp.csc_matrix(W_res))
                first = (dt, st, float(d.max()) if d.nnz else 0.0)
            del Wi
            bytes_sum += st["gram_bytes"]
            ks_sum += ks
            recs.append({"col_begin": int(b), "seconds": round(dt, 2), "kernel_s": round(ks, 3),
                         "solve_s": round(dt - st["gram_build_ms"] * 1e-3, 2),
                         "G_build_s": round(st["gram_build_ms"] * 1e-3, 2),
                         "row_GBps": round(st["gram_bytes"] / max(ks, 1e-9) / 1e9, 1),
                         "alg_bytes": st["gram_bytes"], "nnzW": int(st["nnzW"])})
        dt, st, dmax = first
        ks = st["kernel_ms"] * 1e-3
        gbps = bytes_sum / max(ks_sum, 1e-9) / 1e9
        traffic = pmc_traffic(args, span, "item_space_step", True)
        solves = [r["solve_s"] for r in recs]
        kernels = [r["kernel_s"] for r in recs]
        g_s = st["gram_build_ms"] * 1e-3
        return {"columns": int(span), "seconds": round(dt, 2), "value": span / dt, "unit": "item-columns/s",
                "G_build_s": round(g_s, 2), "G_build_split_s": gram_split(st), "kernel_s": round(ks, 2),
                "rows_of_G_read": int(st["gram_rows"]), "kernel": KERNEL_NAMES.get(st["kernel"], st["kernel"]),
                "chosen_by": "SLIMGPU_KERNEL_AUTO (the engine's default)",
                "max_abs_dW_vs_the_timed_step": dmax,
                "nnzW": int(st["nnzW"]),
                "steps": recs,
                "solve_s": {"mean": round(sum(solves) / len(solves), 3), "min": min(solves), "max": max(solves)},
                "kernel_s_per_step": {"mean": round(sum(kernels) / len(kernels), 3), "min": min(kernels),
                                      "max": max(kernels)},
                "value_G_resident": span * len(recs) / max(sum(solves), 1e-9),
                "value_amortised": span * len(recs) / max(sum(solves) + g_s, 1e-9),
                "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": gbps / HBM_PEAK_GBS,
                             "traffic": float(traffic) if traffic else None,
                             "achieved_physical": float(traffic) / ks / 1e9 if traffic and ks > 0 else None,
                             "kernel_ms_per_launch": 1e3 * ks_sum / len(recs),
                             "alg_bytes_per_launch": bytes_sum / len(recs),
                             "launches": len(recs),
                             "note": "byte model of the item-space kernel: the bytes of G its updates and "
                                     "warm-start folds stream (SLIMGPU_LastStats.gram_bytes) over the solver's "
                                     "HIP-event time (union lists included), summed over the ranges; traffic: PMC "
                                     "bytes of the first range's launch (profiles/pmc_traffic.json, matched by "
                                     "configuration and source hash)"},
                "note": "column ranges of the timed steps, from scratch, on the engine's default path; G = R^T R "
                        "(all 100 000 items) is built inside the first range's `seconds` and reported apart "
                        "(G_build_s) and amortised (value_amortised); not `value`: SURVEY.md 8(d) prices the "
                        "residual kernel's traffic"}
    except Exception as e:   # noqa: BLE001 -- an extra must not cost the line
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v


def item_space_whole_matrix(args, mat, ncols, opts, step_rec):
    """Secondary figure, under its own key: north_star's literal target on one GPU -- every item
    column of the 1M x 100K matrix solved once, on the engine's default path (item space; G is on
    the handle: item_space_step paid for it, and `seconds_with_G` adds that cost back).  Its
    roofline object is the one launch's bytes of G over its HIP-event time."""
    saved = {k: os.environ.pop(k, None) for k in ("SLIM_GPU_NO_GRAM", "SLIM_GPU_NO_GRAMCD")}
    try:
        t0 = time.perf_counter()
        Ww, st = mat.learn(col_begin=0, col_end=ncols, **dict(opts, kernel=0))
        dt = time.perf_counter() - t0
        nnzw = int(Ww.nnz)
        del Ww
        ks = st["kernel_ms"] * 1e-3
        gbps = st["gram_bytes"] / max(ks, 1e-9) / 1e9
        g_s = st["gram_build_ms"] * 1e-3 or (step_rec or {}).get("G_build_s", 0.0)
        traffic = pmc_traffic(args, ncols, "item_space_whole_matrix", True)
        return {"columns": int(ncols), "seconds": round(dt, 2), "value": ncols / dt, "unit": "item-columns/s",
                "seconds_with_G": round(dt + (0.0 if st["gram_build_ms"] else g_s), 2),
                "value_with_G": ncols / (dt + (0.0 if st["gram_build_ms"] else g_s)),
                "G_build_s": round(g_s, 2), "kernel_s": round(ks, 2),
                "host_s": round(dt - ks - st["gram_build_ms"] * 1e-3, 2),
                "rows_of_G_read": int(st["gram_rows"]), "nnzW": nnzw,
                "kernel": KERNEL_NAMES.get(st["kernel"], st["kernel"]),
                "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": gbps / HBM_PEAK_GBS,
                             "traffic": float(traffic) if traffic else None,
                             "achieved_physical": float(traffic) / ks / 1e9 if traffic and ks > 0 else None,
                             "kernel_ms_per_launch": st["kernel_ms"], "alg_bytes_per_launch": st["gram_bytes"]},
                "note": "all item columns of the matrix in one SLIMGPU_Learn on the engine's default path "
                        "(north_star's target at N = 1); host_s = D2H + assembly of the model"}
    except Exception as e:   # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v


def item_space_grid(args, dev, npairs=3):
    """Secondary figure, under its own key (never `value`, never priced by SURVEY.md 8(d)'s
    formula): BASELINE.json configs[4] -- synthetic 10M x 20K, ~1e9 nnz, the (l1, l2) pairs of
    test/l12file in file order, R resident, each pair warm-started from the previous model
    (slim_mselect.c:94-113) -- on the path the engine takes for a grid: item-space CD on
    G = R^T R (cd_gram.hpp).  The first `npairs` pairs: the cold pair (pays for G) and one-sweep
    l2 steps (40 of the file's 45 pairs are such steps)."""
    import ctypes as C
    import torch
    from slim_amd import synth
    from slim_amd.engine import KERNEL_AUTO, DeviceMatrix
    saved = {k: os.environ.pop(k, None) for k in ("SLIM_GPU_NO_GRAM", "SLIM_GPU_NO_GRAMCD")}
    try:
        nrows, ncols, target = synth.CONFIGS["c5"]
        rowptr, rowind, _ = synth.generate_csr(nrows, ncols, target, seed=args.seed, device=dev)
        torch.cuda.synchronize()
        mat = DeviceMatrix.from_device_ptrs(nrows, ncols, rowptr.data_ptr(), rowind.data_ptr(), 0,
                                            keepalive=(rowptr, rowind), device=dev.index)
        pairs = [tuple(map(float, ln.split())) for ln in
                 open(os.path.join(ROOT, "tests", "golden", "l12file")) if ln.strip()]
        mat.expect_solves(len(pairs))      # what slim_mselect / Py_SLIM_Mselect announce
        # as Py_SLIM_Mselect / slim_mselect run it (round 6): the models stay in HBM
        # (SLIMGPU_LearnResident), each pair warm-started from the resident previous one; every model
        # is STILL brought to the host here -- the copy is started before the next solve and runs
        # beside it (SLIMGPU_ModelFetchBegin) -- so the figure is comparable with the earlier rounds'
        prev, recs = None, []
        t_all = time.perf_counter()
        for l1, l2 in pairs[:npairs]:
            t0 = time.perf_counter()
            cur, st = mat.learn_resident(warm=prev, l1r=l1, l2r=l2, optTol=1e-7, niters=10000,
                                         seed=args.seed, kernel=KERNEL_AUTO)
            t1 = time.perf_counter()
            if prev is not None:
                h = prev.fetch(return_handle=True)       # begun before this solve
                mat._lib.SLIM_FreeModel(C.byref(C.c_void_p(h)))
                prev.free()
            cur.fetch_begin()
            prev = cur
            dt = time.perf_counter() - t0
            recs.append({"l1": l1, "l2": l2, "seconds": round(dt, 2), "solve_call_s": round(t1 - t0, 2),
                         "kernel_s": round(st["kernel_ms"] * 1e-3, 2),
                         "G_build_s": round(st["gram_build_ms"] * 1e-3, 2),
                         "G_build_split_s": gram_split(st) if st["gram_build_ms"] else None,
                         "kernel": KERNEL_NAMES.get(st["kernel"], st["kernel"]),
                         "sweeps_per_column": round(st["sweeps"] / float(ncols), 2),
                         "rows_of_G_read": int(st["gram_rows"]),
                         "row_GBps": round(st["gram_bytes"] / max(st["kernel_ms"], 1e-9) / 1e6, 1),
                         "nnzW": int(st["nnzW"])})
        t0 = time.perf_counter()
        h = prev.fetch(return_handle=True)
        mat._lib.SLIM_FreeModel(C.byref(C.c_void_p(h)))
        prev.free()
        last_fetch = time.perf_counter() - t0
        total = time.perf_counter() - t_all
        mat.close()
        del rowptr, rowind
        torch.cuda.empty_cache()
        warm = [r["seconds"] for r in recs[1:]]
        last = recs[-1]
        model_gbps = last["row_GBps"]
        return {"workload": "c5 (synthetic %dx%d, ~1e9 nnz, binary), first %d pairs of test/l12file, "
                            "all %d item columns per pair, warm start" % (nrows, ncols, npairs, ncols),
                "pairs": recs, "seconds": round(total, 2),
                "value": npairs * ncols / total, "unit": "item-columns/s",
                "warm_pair_s": round(sum(warm) / len(warm), 2) if warm else None,
                "last_fetch_s": round(last_fetch, 2),
                "models": "resident in HBM, warm start without an upload; every model fetched to the host "
                          "beside the next solve (round 5: host models, 1.31-1.40 s per warm pair)",
                "roofline": {"bound": "hbm", "achieved": model_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": model_gbps / HBM_PEAK_GBS, "traffic": None,
                             "note": "the last pair's launch of the item-space kernel: the bytes of G its "
                                     "updates and warm-start folds streamed (counted on the device: byte "
                                     "planes, SLIMGPU_LastStats.gram_bytes) over its HIP-event time"},
                "byte_model": "bytes of the rows of G read (one per update and per folded warm-start "
                              "coefficient; packed rows: ncols bytes + 8192 per group of ranks that needs "
                              "a second / third byte plane; g stays on chip): row_GBps = that over the "
                              "kernel time",
                "note": "round 3, tile kernel (profiles/r03/c5_grid_45pairs.txt): cold pair 157.9 s, "
                        "one-sweep pairs 38-40 s; the whole 45-pair grid: profiles/r04/"}
    except Exception as e:   # noqa: BLE001 -- an extra must not cost the line
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v


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
# the item-space path on top of those (its launches, G builder and kernels)
GRAM_SOURCES = ("cd_gram.hpp", "cd_gramr.hpp", "gram_pack.hpp", "gram_inst.hpp", "gram_inst.hip",
                "gramr_inst.hpp", "gramr_inst.hip", "gramr_k13.hip")   # (the last three choose the instantiation)


def kernel_hash(kind="tile"):
    """Fingerprint of the solver sources: a PMC figure collected for another build of the
    kernels must not be reported for this one.  kind "gram": the item-space path's sources too."""
    import hashlib
    import re
    h = hashlib.sha256()
    names = KERNEL_SOURCES + (GRAM_SOURCES if kind == "gram" else ())
    for name in names:
        path = os.path.join(ROOT, "slim_amd", "csrc", name)
        h.update(name.encode())     # (a renamed or missing source must not match an old entry)
        with open(path) as f:
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
                e.get("kernel_hash") == kernel_hash("gram" if kernel.startswith("item_space") else "tile")):
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
    and thread-local-PRNG + Gram-column-aTy mode, each on one thread, on --cpu-threads threads
    (32: where this host's memory system still scales) and on all physical cores; estimate
    phase only (the reference's LearnTmr), setup excluded.  `value` = the thread-local-PRNG +
    Gram mode on --cpu-threads threads over --cpu-columns (512) columns.
    Checker use: the same leg verifies the GPU's columns -- one whole tile of the step against
    the oracle walking that tile in the kernel's visiting order (<= 2e-5), and eight sampled
    columns solved again at optTol 1e-12 on the GPU and by the oracle in its own per-item
    order (<= 2e-5: at a tight tolerance the visiting order no longer matters)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import scipy.sparse as sp
    import slim_oracle as O

    # the whole command has to end inside the driver's 1800 s: what is left of WALL_BUDGET_S
    # bounds this leg (the modes below shrink or are skipped, and say so)
    def left():
        return WALL_BUDGET_S - (time.time() - T_START)

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

    def timed(cols, nthreads, mode, budget=0.0):
        if budget > 0:
            O.set_time_budget(budget)
        Wc, stc, _, _ = O.learn_cd(R, cols=np.sort(cols).astype(np.int32), nthreads=nthreads,
                                   return_stats=True, **kw, **mode)
        return Wc, O.learn_seconds(), stc

    # one column on one thread sizes everything else
    _, t1, _ = timed(pool[:1], 1, gram)
    n1 = int(max(1, min(4, (args.cpu_seconds / 2) // max(t1, 1e-3))))
    res = {}
    _, t, _ = timed(pool[:n1], 1, gram)
    res["gram_localprng_1thread"] = {"value": n1 / t, "columns": n1, "seconds": round(t, 3), "threads": 1}
    _, t, _ = timed(pool[:n1], 1, faithful)
    res["fullscan_rand_1thread"] = {"value": n1 / t, "columns": n1, "seconds": round(t, 3), "threads": 1}

    # the headline mode: --cpu-columns (>= 512) columns, one per thread and round
    # on --cpu-threads threads, timed round by round so that the spread of a shared host is
    # visible: `value` = all columns over all seconds, `rounds` lists every round's rate.
    # (128 concurrent columns thrash the memory system of the 2 x 64-core hosts of this pool:
    # the all-cores entries below.)
    use = max(1, min(args.cpu_threads or cores, cores, span))
    rounds = int(max(1, min(span // use, -(-max(args.cpu_columns, use) // use))))
    parts, t, round_rates, D_full = [], 0.0, [], {}
    done_rounds = 0
    for k in range(rounds):
        # (room is kept for the parity checks and the faithful 32-thread round, ~65 s; the
        # all-core modes come last and shrink or drop out by themselves)
        if k >= 2 and left() < 90 + 1.3 * (t / k):
            break
        cols_k = pool[k * use:(k + 1) * use]
        Wk, tk, stk = timed(cols_k, use, gram)
        parts.append(Wk)
        t += tk
        round_rates.append(round(use / tk, 3))
        for c in cols_k:
            D_full[int(c)] = int(stk["D"][c])
        done_rounds += 1
    sample = pool[:use * done_rounds]
    Wc = parts[0]
    for Wk in parts[1:]:
        Wc = Wc + Wk   # disjoint columns
    srt = sorted(round_rates)
    res["gram_localprng_%dthreads" % use] = {
        "value": sample.size / t, "columns": int(sample.size), "seconds": round(t, 3), "threads": use,
        "rounds": round_rates, "round_min": srt[0], "round_median": srt[len(srt) // 2],
        "round_max": srt[-1],
        "note": None if done_rounds == rounds else
        "%d of %d rounds: the command's wall budget (%d s) was nearly spent" % (done_rounds, rounds, WALL_BUDGET_S)}
    best = res["gram_localprng_%dthreads" % use]
    _, tf, _ = timed(pool[:use], use, faithful)
    res["fullscan_rand_%dthreads" % use] = {"value": use / tf, "columns": use, "seconds": round(tf, 3),
                                            "threads": use}

    # ---- parity of the GPU's columns (the checker's other job) ----------------------------
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
    # (1) one whole tile of the step, visit for visit
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
    # (2) eight columns of the timing sample once more at optTol 1e-12, GPU (its own kernel and
    # order) against the oracle in its per-item order: no order-to-order noise left, so the
    # stated tolerance applies as it is -- no constant fitted to a sample
    tight = np.sort(pool[:8]).astype(np.int32)
    ktight = dict(kw, optTol=1e-12, maxniters=100000)
    Wgt, _ = mat.learn(columns=tight, l1r=opts["l1r"], l2r=opts["l2r"], optTol=1e-12, niters=100000,
                       seed=opts["seed"], kernel=opts.get("kernel", 0))
    Wot = O.learn_cd(R, cols=tight, order=O.ORDER_PERM, seed=opts["seed"], aty=O.ATY_GRAM,
                     nthreads=min(8, cores), **ktight)
    dtt = abs(sp.csc_matrix(Wgt)[:, tight] - Wot[:, tight])
    d_tight = float(dtt.max()) if dtt.nnz else 0.0

    # ---- all physical cores: one column per core, bounded by a time budget --
    # 128 concurrent columns take ~3 minutes per round on these hosts; the sample is bounded
    # instead: every core works on its column for `budget` seconds, and a column that is cut
    # off counts as the fraction D_reached / D_full of a column (D = nnz touched, from the
    # rounds above that solved the same column).
    for key, mode in (("gram_localprng_allcores", gram), ("fullscan_rand_allcores", faithful)):
        budget = min(45.0, (left() - 30.0) / 3.0)
        ncol = min(cores, sample.size)
        if cores <= use or budget < 15.0:
            res[key] = {"value": None, "threads": cores,
                        "note": "skipped: %s" % ("--cpu-threads already covers every core" if cores <= use
                                                 else "the command's wall budget (%d s) was nearly spent" % WALL_BUDGET_S)}
            continue
        cols_a = pool[:ncol]
        _, ta, sta = timed(cols_a, ncol, mode, budget=budget)
        frac = 0.0
        nfin = 0
        for c in cols_a:
            c = int(c)
            if sta["conv"][c] >= 0:
                frac += 1.0
                nfin += 1
            elif sta["conv"][c] == -1 and D_full.get(c, 0) > 0:
                frac += min(1.0, float(sta["D"][c]) / D_full[c])
        res[key] = {"value": frac / ta, "columns": ncol, "columns_finished": nfin,
                    "column_equivalents": round(frac, 2), "seconds": round(ta, 3), "threads": ncol,
                    "note": "one column per core, cut off after %.0f s: unfinished columns count as "
                            "D_reached / D_full" % budget}
    O.cache_setup(False)
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
            "tight_max_abs_dW": d_tight, "tight_columns": [int(c) for c in tight],
            "tight_tolerance": 2e-5,
            "sample_max_abs_dW": d_sample, "sample_max_abs_W": w_max, "sample_worst": worst,
            "parity_ok": bool((d_tile is None or d_tile <= 2e-5) and d_tight <= 2e-5),
            "note": "tile: GPU vs oracle_learn_cd_tile walking the same tile in the kernel's "
                    "visiting order (visit for visit); tight: eight sampled columns at optTol "
                    "1e-12, GPU vs the oracle in its own per-item order; both gate parity_ok at the "
                    "stated 2e-5.  sample_max_abs_dW (GPU in tile order vs the oracle in its own "
                    "order, both stopped at optTol 1e-7) is what two valid visiting orders differ "
                    "by -- reported, not gated: the reference differs from itself by 1.7e-3 on "
                    "ml100k across shuffle seeds (SURVEY.md 8c)",
        },
    }


if __name__ == "__main__":
    main()
