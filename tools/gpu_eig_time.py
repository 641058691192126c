#!/usr/bin/env python3
"""Time of the prep step's eigen-solve: csrc/eigh.hip on the GPU against LAPACK (dsyevr on the wanted
pairs) on the host, on Gram matrices with the spectrum of count data.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wisecondor_amd import wisetools as wt  # noqa: E402

out = {}
for n in [int(v) for v in (sys.argv[1:] or ["100", "192", "300", "600", "1200", "2400"])]:
    rng = np.random.default_rng(n)
    bins = 40 * n
    profile = rng.uniform(0.5, 1.5, bins)
    x = rng.poisson(profile * 200.0 * rng.uniform(0.8, 1.2, (n, 1))).astype(np.float64)
    x /= x.sum(axis=1, keepdims=True)
    x -= x.mean(axis=0)
    g = x @ x.T
    dev = torch.from_numpy(g).cuda()
    wt.sym_eigh_leading(dev, 3)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        vals, vecs = wt.sym_eigh_leading(dev, 3)
    gpu = (time.perf_counter() - t0) / reps
    wt._leading_eigenpairs(g, 3)
    t0 = time.perf_counter()
    for _ in range(reps):
        torch.cuda.synchronize()
        gh = dev.cpu().numpy()                  # (the host route fetches the matrix first)
        hv, hvec = wt._leading_eigenpairs(gh, 3)
    host = (time.perf_counter() - t0) / reps
    err = max(np.abs(np.sign(np.dot(vecs[j], hvec[j])) * vecs[j] - hvec[j]).max() for j in range(3))
    out[str(n)] = {"gpu_ms": gpu * 1e3, "host_lapack_ms": host * 1e3, "max_vector_difference": err,
                   "relative_eigenvalue_difference": float(np.abs(vals - hv).max() / hv[0])}
print(json.dumps(out))
