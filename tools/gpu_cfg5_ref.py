#!/usr/bin/env python3
"""GPU box step of the whole-sample cfg5 golden: build tools/cfg5_case.py's 50 kb reference set and
leave its arrays in the shape of a `newref` output file (wisecondor.py:160-170) in
gpurun_out/cfg5_ref.npz, so that the development container can run the REAL reference's toolTest
(wisecondor.py:174-281) and the CPU oracle on whole 50 kb samples against it
(tools/make_goldens.py --only cfg5whole).  Only inputs of the golden come from here; what the HIP
`test` path itself computes is not stored."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cfg5_case  # noqa: E402
from wisecondor_amd import synth  # noqa: E402
from wisecondor_amd import wisetools as wt  # noqa: E402

t0 = time.time()
case = cfg5_case.build(wt, synth, n_test=1)
ref = case["reference"]
out = dict(binsize=np.float64(cfg5_case.BINSIZE), indexes=ref.indexes, distances=ref.distances,
           chromosome_sizes=case["chrom_bins"], mask=ref.mask.astype(bool), masked_sizes=case["masked_bins"],
           pca_mean=ref.pca_mean, pca_components=ref.pca_components,
           cutoff=np.float64(ref.cutoff), threshold=np.float64(case["threshold"]),
           distances_sha256=np.array(hashlib.sha256(ref.distances.tobytes()).hexdigest()))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
path = os.path.join(ROOT, "gpurun_out", "cfg5_ref.npz")
np.savez_compressed(path, **out)
size = os.path.getsize(path)
if size > 58 << 20:
    # too large to travel in one piece: the distances go as float32 hi + float32 lo residual is not
    # exact, so ship the corrected matrix instead and let the container recompute the distances
    # (checked against distances_sha256 there)
    del out["distances"]
    out["corrected"] = np.asarray(case["corrected"])
    np.savez_compressed(path, **out)
    size = os.path.getsize(path)
print("cfg5 reference written: %d bins, %.1f MB, cutoff %r, threshold %r, %.1f s"
      % (ref.n_bins, size / 1e6, ref.cutoff, case["threshold"], time.time() - t0))
