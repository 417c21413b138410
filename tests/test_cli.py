"""The command-line programs (slim_amd/bin/slim_learn, slim_predict, slim_mselect): options
and file formats of the reference's src/programs/*.c, built on the public C ABI only."""
import os
import re
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

import slim_oracle as O
from conftest import GOLDEN, ROOT, has_gpu
from slim_amd.io import read_csr_text, write_csr_text

BIN = os.path.join(ROOT, "slim_amd", "bin")


def run(prog, *args, env=None, check=True):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([os.path.join(BIN, prog)] + [str(a) for a in args], capture_output=True,
                       text=True, env=e, timeout=600)
    if check:
        assert p.returncode == 0, p.stdout + p.stderr
    return p


@pytest.fixture(scope="module", autouse=True)
def _programs():
    if not os.path.exists(os.path.join(BIN, "slim_learn")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "slim_amd", "csrc")])


def test_usage_and_bad_options():
    for prog in ("slim_learn", "slim_predict", "slim_mselect"):
        assert "Usage: %s" % prog in run(prog, "-help").stdout
        assert "Usage: %s" % prog in run(prog).stdout            # no arguments: short help
        bad = run(prog, "-nosuchoption", "x", check=False)
        assert bad.returncode != 0 and "Illegal command-line option" in bad.stderr
    missing = run("slim_learn", "/nonexistent.csr", check=False)
    assert missing.returncode != 0 and "does not exist" in missing.stderr
    assert run("slim_learn", "-simtype=euclid", os.path.join(GOLDEN, "l12file"),
               check=False).returncode != 0


def test_predict_evaluates_like_the_oracle(tmp_path, ml100k):
    """slim_predict model old test (host scorer forced): HR/ARHR of slim_predict.c:181-243."""
    R, T = ml100k
    W = O.learn_cd(R, order=O.ORDER_PERM, aty=O.ATY_GRAM, nthreads=8)
    mdl = str(tmp_path / "ml.model")
    write_csr_text(mdl, sp.csr_matrix(W))
    out = str(tmp_path / "recs.txt")
    p = run("slim_predict", "-nrcmds=10", "-outfile=" + out, mdl,
            os.path.join(GOLDEN, "ml100k-train.csr"), os.path.join(GOLDEN, "ml100k-test.csr"),
            env={"SLIM_PREDICT": "cpu"})
    ev = O.evaluate(W, R, T)
    m = re.search(r"hr: (\S+) hr_head: (\S+) hr_tail: (\S+) arhr: (\S+)", p.stdout)
    assert m, p.stdout
    assert [float(x) for x in m.groups()] == [float("%.4f" % ev[k]) for k in
                                              ("hr", "hr_head", "hr_tail", "arhr")]
    assert "nvalid: 934 nvalid_head: %d nvalid_tail: %d" % (ev["nvalid_head"], ev["nvalid_tail"]) \
        in p.stdout
    ids, _ = O.predict(W, R, 10)
    lines = open(out).read().splitlines()
    assert len(lines) == 934
    assert [int(t) for t in lines[5].split()[0::2]] == ids[5].tolist()
    # -binarize drops the ratings of the history (all 1.0 here: same lists)
    p2 = run("slim_predict", "-binarize", mdl, os.path.join(GOLDEN, "ml100k-train.csr"),
             os.path.join(GOLDEN, "ml100k-test.csr"), env={"SLIM_PREDICT": "cpu"})
    assert re.search(r"hr: (\S+)", p2.stdout).group(1) == m.group(1)


def test_formats_round_trip(tmp_path, automotive):
    """ijv / cluto / csrnv readers feed the same matrix to the library."""
    R, T, _, _ = automotive
    W = O.learn_cd(R, maxniters=20, order=O.ORDER_PERM, aty=O.ATY_GRAM, nthreads=8)
    Wr, Rr, Tr = sp.csr_matrix(W), sp.csr_matrix(R), sp.csr_matrix(T)

    def dump(M, path, fmt):
        with open(path, "w") as f:
            if fmt == "cluto":
                f.write("%d %d %d\n" % (M.shape[0], M.shape[1], M.nnz))
            for r in range(M.shape[0]):
                lo, hi = M.indptr[r], M.indptr[r + 1]
                if fmt == "ijv":
                    for c, v in zip(M.indices[lo:hi], M.data[lo:hi]):
                        f.write("%d %d %.9g\n" % (r, c, v))
                else:
                    off = 1 if fmt == "cluto" else 0
                    f.write(" ".join("%d %.9g" % (c + off, v) for c, v in
                                     zip(M.indices[lo:hi], M.data[lo:hi])) + "\n")
    results = {}
    for fmt in ("csr", "cluto"):
        paths = [str(tmp_path / ("%s.%s" % (n, fmt))) for n in ("w", "r", "t")]
        for M, pth in zip((Wr, Rr, Tr), paths):
            dump(M, pth, fmt)
        p = run("slim_predict", "-ifmt=" + fmt, *paths, env={"SLIM_PREDICT": "cpu"})
        results[fmt] = re.search(r"hr: \S+ hr_head: \S+ hr_tail: \S+ arhr: \S+", p.stdout).group(0)
    ev = O.evaluate(W, R, T)
    assert results["csr"] == results["cluto"]
    assert results["csr"].startswith("hr: %.4f" % ev["hr"])


@pytest.mark.gpu
def test_learn_then_predict_ml100k(tmp_path):
    """README.md:119,131 of the reference: slim_learn then slim_predict on ml100k."""
    mdl = str(tmp_path / "slim.model")
    p = run("slim_learn", "-l1r=1", "-l2r=1", "-dbglvl=3", os.path.join(GOLDEN, "ml100k-train.csr"),
            mdl)
    assert "nrows: 934, ncols: 1683, nnz: 98222" in p.stdout and "Done estimation" in p.stdout
    W = read_csr_text(mdl)
    assert abs(W.nnz - 65928) <= 60
    q = run("slim_predict", mdl, os.path.join(GOLDEN, "ml100k-train.csr"),
            os.path.join(GOLDEN, "ml100k-test.csr"))
    assert "hr: 0.3191" in q.stdout and "arhr: 0.1504" in q.stdout
    # warm start from the model just written, binarized input, fSLIM
    p2 = run("slim_learn", "-ipmdlfile=" + mdl, "-binarize", "-l1r=1", "-l2r=2",
             os.path.join(GOLDEN, "ml100k-train.csr"), str(tmp_path / "m2.model"))
    assert "binarize: Yes" in p2.stdout
    p3 = run("slim_learn", "-nnbrs=20", "-simtype=jac", os.path.join(GOLDEN, "ml100k-train.csr"),
             str(tmp_path / "m3.model"))
    W3 = read_csr_text(str(tmp_path / "m3.model"))
    assert sp.csc_matrix(W3).getnnz(axis=0).max() <= 20 and "simtype: jac, nnbrs: 20" in p3.stdout


@pytest.mark.gpu
def test_mselect_l12file(tmp_path):
    """slim_mselect over a few pairs of the reference's test/l12file, models written per pair."""
    l12 = str(tmp_path / "l12")
    open(l12, "w").write("1.0 1.0\n2.0 1.0\n4.0 5.0\n")
    p = subprocess.run([os.path.join(BIN, "slim_mselect"), os.path.join(GOLDEN, "ml100k-train.csr"),
                        os.path.join(GOLDEN, "ml100k-test.csr"), l12], capture_output=True,
                       text=True, cwd=str(tmp_path), timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    rows = re.findall(r"l1r: (\S+) l2r: (\S+) nnz:\s+(\d+) hr: (\S+)", p.stdout)
    assert len(rows) == 3 and rows[0][3] == "0.3191"
    assert os.path.exists(str(tmp_path / "1.0 1.0.model"))   # slim_mselect.c:110-112
    nnz = [int(r[2]) for r in rows]
    assert nnz[0] > nnz[1] > nnz[2]                          # more l1 => sparser model


@pytest.mark.gpu
def test_learn_admm_then_predict_ml100k(tmp_path):
    """slim_learn -algo=admm (the reference needs an MKL build for it, cmdline_learn.c:25): the
    model scores like the figures of profiles/r02/admm_ml100k.txt."""
    mdl = str(tmp_path / "admm.model")
    p = run("slim_learn", "-algo=admm", "-l1r=1", "-l2r=1", os.path.join(GOLDEN, "ml100k-train.csr"), mdl)
    assert "solver: admm" in p.stdout and "Learning the model using ADMM" in p.stdout
    W = read_csr_text(mdl)
    assert W.shape[0] == 1683 and W.nnz > 100000 and W.data.min() > 0
    q = run("slim_predict", mdl, os.path.join(GOLDEN, "ml100k-train.csr"),
            os.path.join(GOLDEN, "ml100k-test.csr"))
    hr = float(re.search(r"hr: (\S+)", q.stdout).group(1))
    assert 0.31 <= hr <= 0.34                                     # 0.3266 measured


@pytest.mark.gpu
def test_mselect_with_resident_models_prints_the_same_lines(tmp_path):
    """slim_mselect keeps the grid's models in HBM (SLIMGPU_LearnResident / ModelPredict); with
    SLIM_GPU_RESIDENT=0 every model is assembled on the host as before: the same lines (nnz, HR,
    ARHR per pair), the same model files."""
    l12 = str(tmp_path / "l12")
    open(l12, "w").write("1.0 1.0\n2.0 1.0\n4.0 5.0\n0.5 5.0\n")
    outs = []
    for res in ("1", "0"):
        d = tmp_path / ("res" + res)
        d.mkdir()
        env = dict(os.environ, SLIM_GPU_RESIDENT=res)
        p = subprocess.run([os.path.join(BIN, "slim_mselect"), os.path.join(GOLDEN, "ml100k-train.csr"),
                            os.path.join(GOLDEN, "ml100k-test.csr"), l12], capture_output=True,
                           text=True, cwd=str(d), timeout=600, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        outs.append(re.findall(r"(l1r: \S+ l2r: \S+ nnz:\s+\d+ hr: \S+ hr_head: \S+ hr_tail: \S+ arhr: \S+)", p.stdout))
        assert len(outs[-1]) == 4
    assert outs[0] == outs[1]
    for name in ("1.0 1.0.model", "0.5 5.0.model"):
        a = open(str(tmp_path / "res1" / name)).read()
        assert a == open(str(tmp_path / "res0" / name)).read() and len(a) > 1000
