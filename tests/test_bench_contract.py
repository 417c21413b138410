"""bench.py's contract pieces that can be checked without a GPU: the default configuration is
the one BASELINE.json's metric is quoted on, and the PMC traffic figure committed under
profiles/ belongs to THIS build of the kernels (bench.py reports `roofline.traffic` only when
configuration, seed and the hash of the kernel sources all match -- a kernel edited after the
counters were collected must show up here, not as a silent `traffic: null` in the driver's line)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _default_args(**over):
    a = argparse.Namespace(workload="c4", scale=1.0, seed=1)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_default_step_is_the_named_configuration():
    assert bench.DEFAULT_BATCH == 8192 and bench.HBM_PEAK_GBS == 8000.0
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "item-columns solved/sec" in base["metric"]
    from slim_amd import synth
    assert synth.CONFIGS["c4"][:2] == (1000000, 100000) and synth.CONFIGS["c5"][:2] == (10000000, 20000)


def test_committed_pmc_traffic_belongs_to_this_build():
    t = bench.pmc_traffic(_default_args(), bench.DEFAULT_BATCH, "tile32", True)
    assert t is not None and 2.5e14 < float(t) < 3.5e14       # bytes per launch of the default step
    t5 = bench.pmc_traffic(_default_args(workload="c5"), 4096, "tile32", True)
    assert t5 is not None and 1.5e14 < float(t5) < 2.5e14
    # the engine's default path on the default workload (item space) has its own entry, matched by
    # the hash of its sources on top of the residual kernel's
    ti = bench.pmc_traffic(_default_args(), bench.DEFAULT_BATCH, "item_space_step", True)
    assert ti is not None and 1e13 < float(ti) < 6e13
    # (the whole-matrix step of the residual kernel -- 2 x 10 minutes of counters per collection, a
    # path no default takes any more -- is no longer re-collected every round: profiles/r04 has it)
    # another seed, another kernel or a multi-GPU line has no entry: null, never a stale figure
    assert bench.pmc_traffic(_default_args(seed=2), bench.DEFAULT_BATCH, "tile32", True) is None
    assert bench.pmc_traffic(_default_args(), bench.DEFAULT_BATCH, "tile32", True, world=2) is None


def test_kernel_hash_ignores_comments_and_layout(tmp_path, monkeypatch):
    h = bench.kernel_hash()
    assert len(h) == 16 and h == bench.kernel_hash()
    entries = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["entries"]
    assert any(e.get("kernel_hash") == h for e in entries)


def test_dry_run_projection_adds_up():
    """bench.py --dry-run-world N on a stand-in matrix: every shard of the step
    is solved once after one warm-up, the spread is (max - min) / mean of the kernel times, and the
    projected command = start-up + broadcast + warm-up + 20 steps at the slowest shard's wall
    time + the whole-matrix step, held against the driver's 1800 s."""
    calls = []

    class Mat(object):
        nrows = 1000000

        def learn(self, col_begin, col_end, shard, **kw):
            calls.append((col_begin, col_end, shard))
            r = shard[0]
            return None, {"ncols_solved": (col_end - col_begin) // shard[1], "kernel_ms": 50000.0 + 500.0 * r,
                          "alg_bytes": 3.6e14}

    args = _default_args(dry_run_world=8, no_item_space=True)
    out = bench.dry_run(args, Mat(), 100000, 8192, {}, 10 ** 9)
    assert calls[0] == (0, 2048, (0, 8))                       # the warm-up: 1/32 of the range
    assert [c[2] for c in calls[1:]] == [(r, 8) for r in range(8)] and calls[1][:2] == (0, 65536)
    assert out["dry_run_world"] == 8 and [x["rank"] for x in out["ranks"]] == list(range(8))
    assert abs(out["kernel_ms_spread"] - 3500.0 / 51750.0) < 1e-3
    p = out["projected_command_s"]
    assert abs(p["total"] - (p["start_up_and_generate"] + p["broadcast_R"] + p["warmup"] + p["timed_steps"]
                             + p["whole_matrix_step"] + p["whole_matrix_step_item_space"])) < 0.2
    assert p["limit"] == 1800.0 and isinstance(p["fits"], bool)
