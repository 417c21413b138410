#!/usr/bin/env python3
"""Top-N scorer at C4 scale: SLIMGPU_Predict for U users of the synthetic
C4 matrix against a synthetic model of C4's shape (100K x 100K, 2700 entries per row -- the
learned C4 model has 2727 per column).  Prints users/s and checks a few users against the
host scorer (SLIM_GetTopN).

  python scripts/topn_scale.py [--users 65536] [--row-nnz 2700]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=65536)
    ap.add_argument("--row-nnz", type=int, default=2700)
    ap.add_argument("--nrcmds", type=int, default=10)
    args = ap.parse_args()
    import numpy as np
    import torch
    from slim_amd import _lib, synth
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    nrows, ncols, target = synth.CONFIGS["c4"]
    rowptr, rowind, _ = synth.generate_csr(nrows, ncols, target, seed=1, device=dev)
    U = args.users
    hptr = rowptr[:U + 1].cpu().numpy().astype(np.intp)
    hind = rowind[:int(hptr[-1])].cpu().numpy()
    hval = np.ones(hind.size, np.float32)
    del rowptr, rowind
    torch.cuda.empty_cache()
    n, k = ncols, args.row_nnz
    rng = np.random.default_rng(5)
    wind = ((np.arange(n, dtype=np.int64)[:, None] * 7 + np.arange(k, dtype=np.int64)[None, :] * 37) % n)
    wind = np.sort(wind, axis=1).astype(np.int32).ravel()
    wval = (rng.random(n * k, dtype=np.float32) * 0.01).astype(np.float32)
    wptr = (np.arange(n + 1, dtype=np.int64) * k).astype(np.intp)
    hW, hH = C.c_void_p(), C.c_void_p()
    lib.Py_csr_wrapper(n, wptr, wind, wval.ctypes.data_as(C.c_void_p), C.byref(hW))
    lib.Py_csr_wrapper(U, hptr, hind, hval.ctypes.data_as(C.c_void_p), C.byref(hH))
    N = args.nrcmds
    out = np.full(U * N, -1, np.int32)
    sc = np.zeros(U * N, np.float32)
    for rep in range(2):
        t0 = time.time()
        rc = lib.SLIMGPU_Predict(N, hW, hH, out, sc)
        dt = time.time() - t0
        assert rc == 1, _lib.last_error()
        adds = float(np.sum(np.diff(hptr))) * k
        print(json.dumps({"users": U, "history_nnz": int(hind.size), "model_nnz": int(wind.size),
                          "seconds_incl_h2d": round(dt, 3), "users_per_s": round(U / dt, 1),
                          "scatter_adds_per_s": round(adds / dt / 1e9, 2), "unit": "1e9/s", "rep": rep}),
              flush=True)
    # the one-wavefront-per-user kernel (vectors in HBM) on the first 2048 users: same lists?
    U2 = min(U, 2048)
    hH2 = C.c_void_p()
    lib.Py_csr_wrapper(U2, np.ascontiguousarray(hptr[:U2 + 1]), hind, hval.ctypes.data_as(C.c_void_p), C.byref(hH2))
    out2 = np.full(U2 * N, -1, np.int32)
    sc2 = np.zeros(U2 * N, np.float32)
    os.environ["SLIM_TOPN_KERNEL"] = "wave"
    t0 = time.time()
    assert lib.SLIMGPU_Predict(N, hW, hH2, out2, sc2) == 1
    dt2 = time.time() - t0
    del os.environ["SLIM_TOPN_KERNEL"]
    print(json.dumps({"wave_kernel_users": U2, "seconds_incl_h2d": round(dt2, 3),
                      "chunk_equals_wave_kernel": bool(np.array_equal(out2, out[:U2 * N]) and
                                                       np.array_equal(sc2, sc[:U2 * N]))}), flush=True)
    # host scorer on a few users
    iopt = np.full(40, -1, np.int32)
    same = True
    for u in (0, 1, U // 2, U - 1):
        s, e = int(hptr[u]), int(hptr[u + 1])
        ids = np.zeros(N, np.int32)
        scs = np.zeros(N, np.float32)
        t0 = time.time()
        hv = np.ascontiguousarray(hval[s:e])
        cnt = lib.SLIM_GetTopN(hW, e - s, np.ascontiguousarray(hind[s:e]), hv.ctypes.data_as(C.c_void_p),
                               iopt.ctypes.data_as(C.c_void_p), N, ids, scs)
        host_s = time.time() - t0
        same &= bool(np.array_equal(ids[:cnt], out[u * N:u * N + cnt]) and
                     np.array_equal(scs[:cnt], sc[u * N:u * N + cnt]))
    print(json.dumps({"bit_identical_to_host_on_4_users": same, "host_seconds_per_user": round(host_s, 4)}))


if __name__ == "__main__":
    main()
