"""Ad-hoc GPU check of the tile kernel: ml100k vs oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import slim_oracle as O
from slim_amd.engine import DeviceMatrix, KERNEL_TILE, KERNEL_WAVE_LDS
from slim_amd.io import read_csr_text

R = read_csr_text(os.path.join(ROOT, "tests/golden/ml100k-train.csr"))
T = read_csr_text(os.path.join(ROOT, "tests/golden/ml100k-test.csr"))
Wo, so, eo, oo = O.learn_cd(R, order=O.ORDER_PERM, seed=1, aty=O.ATY_GRAM, return_stats=True, nthreads=16)
mat = DeviceMatrix.from_scipy(R)
for kern in (KERNEL_WAVE_LDS, KERNEL_TILE):
    for rep in range(2):
        W, st = mat.learn(seed=1, kernel=kern)
    d = abs(W - Wo)
    cs = mat.column_stats()
    print("kernel", kern, "ms %.3f nnz %d (oracle %d) max|dW| %.3e loss %.5e fit %.5e  sweeps sum %d (oracle %d) na equal %s G equal %s"
          % (st["kernel_ms"], W.nnz, Wo.nnz, d.max(), st["objval"], st["error"], cs.sweeps.sum(), so["sweeps"].sum(),
             bool((cs.nacols == so["nacols"]).all()), bool((cs.G == so["G"]).all())))
    ev = O.evaluate(W, R, T)
    print("  HR %.4f ARHR %.4f" % (ev["hr"], ev["arhr"]))
Wt, _ = mat.learn(seed=1, kernel=KERNEL_TILE, optTol=1e-12, niters=100000)
Wr = O.learn_cd(R, order=O.ORDER_PERM, seed=1, aty=O.ATY_GRAM, nthreads=16, optTol=1e-12, maxniters=100000)
print("tight: max|dW| %.3e" % abs(Wt - Wr).max())
