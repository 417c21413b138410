"""Host-side mirror of the reference's Python package (slim_amd.SLIM / SLIMatrix):
parameter handling, id mapping, marshalling -- the parts that run without a GPU --
and, in the build container only, the reference's own wrapper loaded against this
repo's libslim.so to confirm the Py_* ABI."""
import ctypes as C
import os
import sys
import types

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT
from slim_amd import SLIM, SLIMatrix, _lib
from slim_amd.constants import SLIM_NOPTIONS, Opt
from slim_amd.interface import build_options, check_params


def test_defaults_match_reference_wrapper():
    # core.py:123-198: niters 50 (not the C default 10000), nrcmds 10, optTol 1e-7 ...
    p = {}
    io, do = build_options(check_params(p))
    assert p == dict(dbglvl=0, nnbrs=0, simtype="cos", algo="cd", nthreads=1, niters=50,
                     nrcmds=10, l1r=1.0, l2r=1.0, optTol=1e-7, ordered=0)
    want_i = np.full(SLIM_NOPTIONS, -1, np.int32)
    want_i[[Opt.DBGLVL, Opt.NNBRS, Opt.SIMTYPE, Opt.NTHREADS, Opt.MAXNITERS, Opt.ALGO,
            Opt.ORDERED, Opt.NRCMDS]] = [0, 0, 0, 1, 50, 1, 0, 10]
    want_d = np.full(SLIM_NOPTIONS, -1.0)
    want_d[[Opt.L1R, Opt.L2R, Opt.OPTTOL]] = [1.0, 1.0, 1e-7]
    assert np.array_equal(io, want_i) and np.array_equal(do, want_d)
    # README params (README.md:124)
    io, do = build_options(check_params({"algo": "cd", "nthreads": 2, "l1r": 1.0, "l2r": 1.0}))
    assert io[Opt.NTHREADS] == 2 and io[Opt.MAXNITERS] == 50


def test_attribute_bag_params():
    ns = types.SimpleNamespace(l1r=2.0, niters=7)
    io, do = build_options(check_params(ns))
    assert ns.nrcmds == 10 and ns.ordered == 0
    assert io[Opt.MAXNITERS] == 7 and do[Opt.L1R] == 2.0


@pytest.mark.parametrize("bad", [{"dbglvl": -1}, {"nnbrs": 1.5}, {"simtype": "cosine"},
                                 {"algo": "sgd"}, {"nthreads": 0}, {"niters": 0},
                                 {"nrcmds": "10"}, {"l1r": -0.1}, {"l2r": "x"}, {"optTol": -1}])
def test_bad_params_raise_typeerror(bad):
    with pytest.raises(TypeError):
        check_params(dict(bad))


def test_fslim_forces_cd(capsys):
    p = {"nnbrs": 5, "algo": "admm"}
    check_params(p)
    assert p["algo"] == "cd" and "fSLIM" in capsys.readouterr().out


def test_id_mapping_first_appearance(automotive_triplets, automotive):
    trn, tst = automotive_triplets
    R, T, users, items = automotive
    m = SLIMatrix(trn)
    assert (m.nUsers, m.nItems) == (2928, 1835)
    assert np.array_equal(m.id2user, users) and np.array_equal(m.id2item, items)
    assert m.id2user[:3].tolist() == [5.0, 6.0, 8.0]  # first rows of AutomotiveTrain.ijv
    view = C.cast(m.handle, C.POINTER(_lib.CsrView)).contents
    assert view.nrows == 2928 and view.ncols == 1835
    nnz = int(view.rowptr[2928])
    assert nnz == 17545
    assert np.array_equal(np.ctypeslib.as_array(view.rowind, shape=(nnz,)), R.indices)
    assert np.array_equal(np.ctypeslib.as_array(view.rowval, shape=(nnz,)),
                          R.data.astype(np.float32))
    # a second matrix reuses the maps; unknown users/items are dropped with a message
    v = SLIMatrix(tst, m)
    assert (v.nUsers, v.nItems) == (2928, 1835)
    extra = np.vstack([tst[:5], [[1e9, 1e9, 1.0]]])
    v2 = SLIMatrix(extra, m)
    assert int(C.cast(v2.handle, C.POINTER(_lib.CsrView)).contents.rowptr[2928]) == 5


def test_list_and_dataframe_inputs(automotive_triplets):
    trn, _ = automotive_triplets
    a = SLIMatrix(trn[:500])
    b = SLIMatrix([[str(int(u)), str(int(i)), float(v)] for u, i, v in trn[:500]])
    assert a.nUsers == b.nUsers and a.nItems == b.nItems
    assert b.id2item.dtype.kind == "U"  # raw ids are kept as given
    pd = pytest.importorskip("pandas")
    c = SLIMatrix(pd.DataFrame(trn[:500]))
    assert np.array_equal(c.id2item, a.id2item)
    with pytest.raises(TypeError):
        SLIMatrix("not a matrix")


def test_csr_input_and_oldmat_shape_check(ml100k):
    R, _ = ml100k
    m = SLIMatrix(R)
    assert (m.nUsers, m.nItems) == R.shape
    assert np.array_equal(m.id2item, np.arange(1683))
    with pytest.raises(TypeError):
        SLIMatrix(sp.csr_matrix((3, 4)), m)


def test_untrained_model_errors(ml100k):
    R, _ = ml100k
    s = SLIM()
    with pytest.raises(TypeError):
        s.predict(SLIMatrix(R))
    with pytest.raises(RuntimeError):
        s.save_model("a", "b")
    with pytest.raises(RuntimeError):
        s.to_csr()
    with pytest.raises(RuntimeError):
        s.load_model("/nonexistent", "/nonexistent")


def test_load_predict_export_without_gpu(tmp_path, ml100k):
    """save/load/predict/to_csr are host paths: exercise them with an oracle-made model."""
    import slim_oracle as O
    from slim_amd.io import write_csr_text
    R, _ = ml100k
    cols = np.arange(0, 1683, 3, dtype=np.int32)
    W = O.learn_cd(R, order=O.ORDER_PERM, aty=O.ATY_GRAM, cols=cols, nthreads=4)
    write_csr_text(str(tmp_path / "m.csr"), sp.csr_matrix(W))
    np.savetxt(str(tmp_path / "map.csv"), np.arange(1683), fmt="%s")
    s = SLIM()
    s.load_model(str(tmp_path / "m.csr"), str(tmp_path / "map.csv"))
    assert s.nItems == 1683
    back = s.to_csr()
    assert abs(back - sp.csr_matrix(W)).max() == 0
    m = SLIMatrix(R)
    out, sc = s.predict(m, nrcmds=10, returnscores=True, outfile=str(tmp_path / "o.txt"))
    ids, scores = O.predict(W, R, 10)
    assert len(out) == 934
    for u in (0, 17, 933):
        filled = ids[u] >= 0
        assert np.array_equal(out[u][filled], ids[u][filled])
        assert np.array_equal(sc[u], scores[u])
    assert os.path.getsize(str(tmp_path / "o.txt")) > 0
    s.save_model(str(tmp_path / "m2.csr"), str(tmp_path / "map2.csv"))
    s2 = SLIM()
    s2.load_model(str(tmp_path / "m2.csr"), str(tmp_path / "map2.csv"))
    assert abs(s2.to_csr() - back).max() == 0


@pytest.mark.skipif(not os.path.isdir("/root/reference/python-package/SLIM"),
                    reason="the reference tree exists only in the build container")
def test_reference_python_wrapper_binds_to_this_library(tmp_path, monkeypatch, ml100k):
    """Import the reference's unmodified Python package with its libslim.so lookup
    (site-packages/SLIM/libslim.so, core.py:31-43) pointed at this repo's library and
    drive the Py_* entry points that need no GPU."""
    import site
    fake = tmp_path / "site"
    (fake / "SLIM").mkdir(parents=True)
    os.symlink(_lib.LIB_PATH, str(fake / "SLIM" / "libslim.so"))
    monkeypatch.setattr(site, "getsitepackages", lambda: [str(fake)])
    monkeypatch.syspath_prepend("/root/reference/python-package")
    monkeypatch.setattr(sp, "csr", types.SimpleNamespace(csr_matrix=sp.csr_matrix), raising=False)
    for k in [k for k in sys.modules if k == "SLIM" or k.startswith("SLIM.")]:
        monkeypatch.delitem(sys.modules, k)
    ref = pytest.importorskip("SLIM")
    R, _ = ml100k
    mat = ref.SLIMatrix(R)
    assert mat.nUsers == 934
    import slim_oracle as O
    from slim_amd.io import write_csr_text
    W = O.learn_cd(R, order=O.ORDER_PERM, aty=O.ATY_GRAM, cols=np.arange(0, 1683, 4, dtype=np.int32),
                   nthreads=4)
    write_csr_text(str(tmp_path / "m.csr"), sp.csr_matrix(W))
    np.savetxt(str(tmp_path / "map.csv"), np.arange(1683), fmt="%s")
    model = ref.SLIM()
    model.load_model(str(tmp_path / "m.csr"), str(tmp_path / "map.csv"))
    out = model.predict(mat, nrcmds=10)
    ids, _ = O.predict(W, R, 10)
    filled = ids[7] >= 0
    assert np.array_equal(np.asarray(out[7])[filled], ids[7][filled])
    assert abs(model.to_csr() - sp.csr_matrix(W)).max() == 0
    # training through the reference wrapper reaches the engine and, without a GPU,
    # fails loudly instead of falling back
    from conftest import has_gpu
    if not has_gpu():
        with pytest.raises(RuntimeError):
            model.train({"niters": 5}, mat)
    for k in [k for k in sys.modules if k == "SLIM" or k.startswith("SLIM.")]:
        sys.modules.pop(k, None)
