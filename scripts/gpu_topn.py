import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp
import slim_oracle as O
from slim_amd import _lib
from slim_amd.engine import _scipy_to_model_handle
from slim_amd.io import read_csr_text
lib = _lib.load()
R = read_csr_text(os.path.join(ROOT, "tests/golden/ml100k-train.csr"))
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = R[:nu].tocsr()
W = O.learn_cd(read_csr_text(os.path.join(ROOT, "tests/golden/ml100k-train.csr")), order=O.ORDER_PERM, aty=O.ATY_GRAM, nthreads=16)
hm = _scipy_to_model_handle(lib, W)
h = C.c_void_p(); val = np.ascontiguousarray(R.data, np.float32)
lib.Py_csr_wrapper(R.shape[0], np.ascontiguousarray(R.indptr, np.intp), np.ascontiguousarray(R.indices, np.int32), val.ctypes.data_as(C.c_void_p), C.byref(h))
n = 10
out = np.full(nu * n, -1, np.int32); sc = np.zeros(nu * n, np.float32)
print("calling", flush=True)
t = time.time(); rc = lib.SLIMGPU_Predict(n, hm, h, out, sc); print("rc", rc, "sec", time.time() - t, flush=True)
ids, scores = O.predict(W, R, n)
print("equal ids", np.array_equal(out.reshape(nu, n), ids), "equal scores", np.array_equal(sc.reshape(nu, n), scores))
print(out.reshape(nu, n)[:2], ids[:2])
