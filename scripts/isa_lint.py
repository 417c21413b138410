#!/usr/bin/env python3
"""ISA lint for the HIP translation units: exec-dependent instructions in the shadow of a partial
EXEC mask.

What it looks for.  Structured control flow on gfx9 ends a divergent region with
`s_or_b64 exec, exec, s[saved]` at the top of the join block.  Everything the compiler puts IN
FRONT of that instruction inside the join block still runs under the region's partial mask.  Scalar
instructions, v_readlane / v_writelane (SGPR spills: they ignore EXEC) and waits are harmless there;
a VGPR spill (`scratch_store` / `scratch_load`), or any other vector instruction, is not: only the
lanes that were active in the region store (or reload) their value, the others keep what the slot
held before.

Round 6 found exactly that in one instantiation of cd_gramr_kernel ("the two-ahead
ring"): LLVM (ROCm 7.2, clang 22) folded a copy of the batch header's row record into a 16-byte
spill store and placed it in the join block of fetch_g's `if (want && in_lds && my_wave)` BEFORE
the `s_or_b64 exec` -- lanes outside that branch later reloaded a stale record, i.e. a wild
hi-plane offset: wrong models or a memory fault, depending on what the slot held.  Whether the
store lands before or behind the restore depends on register allocation (the three-ahead ring of
the same source had it behind), so every build is checked.

usage: isa_lint.py [file.s ...]      device assembly from `hipcc --save-temps` / `-S`
       isa_lint.py --build           compile every .hip of slim_amd/csrc to device assembly
                                     (hipcc --offload-device-only -S, flags of the Makefile) and lint it
exit status 1 if anything is flagged."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "slim_amd", "csrc")

# a VGPR spill or reload: scratch access, or whatever the compiler annotated as one
_SPILL = re.compile(r"^(scratch_(store|load)|buffer_(store|load)\S*\s.*\boffen\b)")
_SPILL_NOTE = re.compile(r";.*\b(Spill|Reload)\b")
_EXEC_FREE = re.compile(r"^(v_readlane_b32|v_writelane_b32)")  # SGPR spills: they ignore EXEC
_RESTORE = re.compile(r"^s_or_b64\s+exec,\s*exec,")
_EXEC_WRITE = re.compile(r"^s_\w+\s+exec\b|^s_(and|or|andn2|xor|orn2)_saveexec_b64|^v_cmpx")
_LABEL = re.compile(r"^([.\w$]+):")
_BRANCH = re.compile(r"^s_(cbranch|branch|endpgm|setpc)")


def lint_text(text, name="<asm>"):
    """-> list of (kernel, line number, instruction, label of the join block): VGPR spills / reloads
    that sit in a join block in front of its `s_or_b64 exec, exec, ...`, with no other write of
    EXEC in between (a region opened and closed inside one block is the region's own code)."""
    findings = []
    kernel = None
    block_label = None
    pending = []
    open_block = False
    # join blocks: the targets of `s_cbranch_execz` -- the skip over a divergent region lands there
    # with no lane active, the region's own code falls into it with the partial mask (a labeled
    # block INSIDE a region, reached by other branches, may reload what the region itself uses)
    joins = set(re.findall(r"^\s*s_cbranch_execz\s+([.\w$]+)", text, re.M))
    for ln, raw in enumerate(text.split("\n"), 1):
        stripped = raw.strip()
        if not stripped or stripped.startswith(";"):
            continue
        line = stripped.split(";")[0].strip()
        if not line:
            continue
        m = _LABEL.match(line)
        if m:
            lab = m.group(1)
            if not lab.startswith(".L") and not lab.startswith("$"):
                kernel = lab
            block_label, pending, open_block = lab, [], lab in joins
            continue
        if line.startswith(".") or not open_block:
            continue
        if _RESTORE.match(line):
            findings += [(kernel, pl, pi, block_label) for (pl, pi) in pending]
            pending = []
            continue  # (a block may restore several nested masks in a row)
        if _BRANCH.match(line) or _EXEC_WRITE.match(line):
            open_block = False
            continue
        if _EXEC_FREE.match(line):
            continue
        if _SPILL.match(line) or _SPILL_NOTE.search(stripped):
            pending.append((ln, line))
    return findings


def lint_file(path):
    with open(path) as f:
        return lint_text(f.read(), path)


def makefile_flags():
    flags = {"CXXFLAGS": "", "HIPFLAGS": "", "ARCH": "gfx950"}
    with open(os.path.join(CSRC, "Makefile")) as f:
        for l in f:
            m = re.match(r"^(CXXFLAGS|HIPFLAGS|ARCH)\s*\??=\s*(.*)$", l)
            if m:
                flags[m.group(1)] = m.group(2).strip()
    hip = flags["HIPFLAGS"].replace("$(ARCH)", flags["ARCH"])
    return flags["CXXFLAGS"].split() + hip.split()


def build_asm(src, outdir, extra=()):
    out = os.path.join(outdir, os.path.basename(src) + ".s")
    cmd = ["/opt/rocm/bin/hipcc"] + makefile_flags() + list(extra) + ["--offload-device-only", "-S", "-o", out, src]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def main(argv):
    files = [a for a in argv if not a.startswith("--")]
    if "--build" in argv:
        outdir = tempfile.mkdtemp(prefix="isa_lint_")
        srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
        with ThreadPoolExecutor(max_workers=min(16, len(srcs))) as ex:
            files += list(ex.map(lambda s: build_asm(s, outdir), srcs))
    bad = 0
    for f in files:
        fs = lint_file(f)
        nk = len(re.findall(r"^\s*\.amdhsa_kernel ", open(f).read(), re.M))
        print("%s: %d kernel(s), %d finding(s)" % (os.path.basename(f), nk, len(fs)))
        for (k, ln, ins, lab) in fs:
            print("  %s:%d  in join block %s of %s: `%s` runs before the exec restore" % (os.path.basename(f), ln, lab, k, ins))
        bad += len(fs)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
