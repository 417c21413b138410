#!/usr/bin/env python3
"""Basic-block summary of one kernel in a `hipcc --save-temps` .s file: per block the number of
instructions, global loads / stores, scratch (spill) traffic, LDS ops, barriers and the branch
targets -- to see which spills sit inside the loops that matter.

  python scripts/isa_blocks.py file.s 'cd_tile_kernelILi32ELb0ELb0ELi16ELb0E' [--min-scratch 1]
"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    only_scratch = "--scratch" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*%s\w*:" % re.escape(pat), l):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = []
    cur = {"name": "entry", "n": 0, "gl": 0, "gs": 0, "sl": 0, "ss": 0, "ds": 0, "bar": 0, "br": [],
           "wl": 0, "rl": 0, "line": start}
    for i in range(start + 1, len(lines)):
        l = lines[i].strip()
        if l.startswith(".Lfunc_end") or l.startswith(".end_amdhsa_kernel"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"name": m.group(1), "n": 0, "gl": 0, "gs": 0, "sl": 0, "ss": 0, "ds": 0, "bar": 0,
                   "br": [], "wl": 0, "rl": 0, "line": i}
            continue
        if not l or l.startswith(";") or l.startswith("."):
            continue
        op = l.split()[0]
        cur["n"] += 1
        if op.startswith("global_load") or op.startswith("buffer_load"):
            cur["gl"] += 1
        elif op.startswith("global_store") or op.startswith("global_atomic"):
            cur["gs"] += 1
        elif op.startswith("scratch_load"):
            cur["sl"] += 1
        elif op.startswith("scratch_store"):
            cur["ss"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
        elif op == "s_barrier":
            cur["bar"] += 1
        elif op == "v_writelane_b32":
            cur["wl"] += 1
        elif op == "v_readlane_b32":
            cur["rl"] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            cur["br"].append(l.split()[-1])
    blocks.append(cur)
    idx = {b["name"]: k for k, b in enumerate(blocks)}
    print("%-12s %6s %5s %4s %4s %4s %4s %4s %4s %4s  %s" %
          ("block", "line", "inst", "gld", "gst", "scL", "scS", "lds", "bar", "wl/rl", "branches (<- = back edge)"))
    for k, b in enumerate(blocks):
        if only_scratch and not (b["sl"] or b["ss"]):
            continue
        br = " ".join(("<-" if idx.get(t, 1 << 30) <= k else "") + t for t in b["br"])
        print("%-12s %6d %5d %4d %4d %4d %4d %4d %4d %2d/%-3d %s" %
              (b["name"], b["line"], b["n"], b["gl"], b["gs"], b["sl"], b["ss"], b["ds"], b["bar"],
               b["wl"], b["rl"], br))


if __name__ == "__main__":
    main()
