"""Ad-hoc GPU check: ml100k full solve vs the oracle, both kernel flavours."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import slim_oracle as O
from slim_amd.engine import DeviceMatrix, KERNEL_WAVE_HBM, KERNEL_WAVE_LDS
from slim_amd.io import read_csr_text

R = read_csr_text(os.path.join(ROOT, "tests/golden/ml100k-train.csr"))
T = read_csr_text(os.path.join(ROOT, "tests/golden/ml100k-test.csr"))
t = time.time(); Wo, so, eo, oo = O.learn_cd(R, order=O.ORDER_PERM, seed=1, aty=O.ATY_GRAM, return_stats=True, nthreads=8)
print("oracle(perm) %.2fs nnz %d loss %.5e fit %.5e" % (time.time() - t, Wo.nnz, oo, eo))
mat = DeviceMatrix.from_scipy(R)
cp, ci, cv, cn = mat.column_view()
Rc = R.tocsc(); Rc.sort_indices()
print("colview ok:", np.array_equal(cp, Rc.indptr), np.array_equal(ci, Rc.indices), np.array_equal(cv, Rc.data))
for kern in (KERNEL_WAVE_LDS, KERNEL_WAVE_HBM):
    for rep in range(2):
        W, st = mat.learn(seed=1, kernel=kern)
    d = abs(W - Wo)
    print("kernel", kern, "ms %.3f total %.3f nnz %d sumW %.6f max|dW| %.3e loss %.5e fit %.5e cols/s %.0f alg GB/s %.1f"
          % (st["kernel_ms"], st["total_ms"], W.nnz, W.data.astype(np.float64).sum(), d.max() if d.nnz else 0,
             st["objval"], st["error"], 1683 / (st["kernel_ms"] / 1e3), st["alg_bytes"] / st["kernel_ms"] / 1e6))
    cs = mat.column_stats()
    print("  D", cs.D.sum(), so["D"].sum(), "U", cs.U.sum(), so["U"].sum(), "G", cs.G.sum(), so["G"].sum(),
          "sweeps equal:", int((cs.sweeps == so["sweeps"]).sum()), "/", len(cs.sweeps), "na equal", bool((cs.nacols == so["nacols"]).all()))
    ev = O.evaluate(W, R, T)
    print("  HR %.4f ARHR %.4f" % (ev["hr"], ev["arhr"]))
