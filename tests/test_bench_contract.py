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
    whole = bench.pmc_traffic(_default_args(), 100000, "tile32", True)
    assert whole is not None
    # another seed, another kernel or a multi-GPU line has no entry: null, never a stale figure
    assert bench.pmc_traffic(_default_args(seed=2), bench.DEFAULT_BATCH, "tile32", True) is None
    assert bench.pmc_traffic(_default_args(), bench.DEFAULT_BATCH, "tile32", True, world=2) is None


def test_kernel_hash_ignores_comments_and_layout(tmp_path, monkeypatch):
    h = bench.kernel_hash()
    assert len(h) == 16 and h == bench.kernel_hash()
    entries = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["entries"]
    assert any(e.get("kernel_hash") == h for e in entries)
