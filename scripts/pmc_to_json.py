#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes of the default bench into profiles/pmc_traffic.json.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d D1 -o c4 -- python bench.py --warmup 0 --steps 1 --cpu-seconds 0
  rocprofv3 --pmc WRITE_SIZE ... -d D2 ...
  rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE ... -- scripts/micro/gather_bw c      (calibration, known bytes)
  python scripts/pmc_to_json.py D1/c4_counter_collection.csv D2/c4_counter_collection.csv \
         CAL_FETCH.csv CAL_WRITE.csv BENCH.json

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM section); the correction factors are taken from the calibration
launches, which move a known number of bytes in the tile kernel's access pattern (random
whole 128-byte lines).
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter(path, kernel_substr, name, take=None):
    """Sum of a counter over the dispatches of a kernel (take: only the first `take` of them in
    time order -- the bench command's extras launch the tile kernel again to build G; take < 0:
    the last -take of them)."""
    rows = [r for r in csv.DictReader(open(path))
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if take is not None:
        rows = rows[:take] if take >= 0 else rows[take:]
    tot = sum(float(r["Counter_Value"]) for r in rows)
    secs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 for r in rows]
    return tot, len(rows), secs


def main():
    fetch_csv, write_csv, cal_f, cal_w, bench_json = sys.argv[1:6]
    known = 256.0 * 16 * 1000 * 16 * 256.0  # gather_bw c: bytes per calibration launch
    gf, _, _ = counter(cal_f, "gather<128", "FETCH_SIZE")
    sw, _, _ = counter(cal_w, "scatter<128", "WRITE_SIZE")
    sf, _, _ = counter(cal_f, "scatter<128", "FETCH_SIZE")
    fcorr, wcorr = known / (gf * 1024), known / (sw * 1024)
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    cfg = bench["config"]
    kname = "cd_tile_kernel" if cfg["kernel"].startswith("tile") else "cd_wave_kernel"
    item_space = os.environ.get("PMC_ITEM_SPACE")   # entry for the item_space_step launch of the same run
    take = int(bench.get("steps", 1))
    if item_space == "whole":   # the whole-matrix launch: the last dispatch of the item-space kernel
        kname, take = "cd_gramr_kernel<10", -1
    elif item_space:
        kname, take = "cd_gramr_kernel<10", 1
    f, nf, tf = counter(fetch_csv, kname, "FETCH_SIZE", take)
    w, nw, tw = counter(write_csv, kname, "WRITE_SIZE", take)
    entry = {
        "match": {"workload": cfg["workload"].split(" ")[0], "scale": cfg["scale"],
                  "seed": cfg.get("seed", 1),
                  "columns_per_step_per_gpu": cfg["columns_per_step_per_gpu"],
                  "kernel": cfg["kernel"], "binary": "binary values" in cfg["workload"]},
        "kernel_hash": bench["roofline"].get("kernel_hash"),
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- "
                   "python bench.py --warmup 0 --steps 1 --cpu-seconds 0 (one pass per counter)",
        "launches": nf, "FETCH_SIZE_kb": f / max(nf, 1), "WRITE_SIZE_kb": w / max(nw, 1),
        "calibration": {"known_bytes_per_launch": known, "gather128_FETCH_SIZE_kb": gf,
                        "scatter128_WRITE_SIZE_kb": sw, "scatter128_FETCH_SIZE_kb": sf},
        "fetch_correction": round(fcorr, 4), "write_correction": round(wcorr, 4),
        "traffic_bytes_per_launch": (f / max(nf, 1)) * 1024 * fcorr + (w / max(nw, 1)) * 1024 * wcorr,
        "kernel_seconds_under_pmc": [round(x, 2) for x in tf + tw],
        "alg_bytes_per_launch_same_run": bench["roofline"]["alg_bytes_per_launch"],
    }
    if item_space:
        # a streaming kernel: FETCH_SIZE tallies its 128-byte requests at 64 bytes like the
        # calibration gathers (MI355X_MICROARCH.md, HBM section); same corrections
        sys.path.insert(0, ROOT)
        import bench as B
        entry["kernel_hash"] = B.kernel_hash("gram")
        if item_space == "whole":
            entry["match"]["kernel"] = "item_space_whole_matrix"
            entry["match"]["columns_per_step_per_gpu"] = bench["item_space_whole_matrix"]["columns"]
            entry["alg_bytes_per_launch_same_run"] = bench["item_space_whole_matrix"]["roofline"]["alg_bytes_per_launch"]
        else:
            entry["match"]["kernel"] = "item_space_step"
            # (the first range's launch: the bytes of that launch, not the mean over the ranges)
            r0 = bench["item_space_step"]["steps"][0]
            entry["alg_bytes_per_launch_same_run"] = r0["alg_bytes"]
    entry["traffic_over_algorithmic"] = entry["traffic_bytes_per_launch"] / max(
        entry["alg_bytes_per_launch_same_run"], 1.0)
    secs = entry["kernel_seconds_under_pmc"]
    if secs:
        entry["physical_GBps_under_pmc"] = entry["traffic_bytes_per_launch"] / (sum(secs) / len(secs)) / 1e9
    out = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        entries = json.load(open(out))["entries"]
    except (OSError, ValueError, KeyError):
        entries = []
    # one entry per configuration: a new collection replaces the old one
    entries = [e for e in entries if e.get("match") != entry["match"]] + [entry]
    json.dump({"entries": entries}, open(out, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
