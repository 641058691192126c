#!/usr/bin/env python3
"""GPU box step of the cfg5 golden: run tools/cfg5_case.py's 50 kb case and leave sample 0's cleaned
z vectors of chromosomes 1-3 (plus what the GPU path called there) in gpurun_out/cfg5_z.npz.
tools/make_goldens.py --only cfg5 then runs the REAL reference's fillTri + segmentTri on them."""
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
case = cfg5_case.build(wt, synth)
zs, rs, xpca, (z, r, n, sd) = cfg5_case.cleaned_regions(wt, case, 0)
outs = wt.test_batch(case["reference"], case["tests"], case["threshold"])
calls0 = np.asarray(outs[0]["results_calls"], dtype=np.float64).reshape(-1, 5)
whole, segs = wt.stouffer_segments([zs[c] for c in cfg5_case.GOLDEN_CHROMS], case["threshold"], 3)
out = {"threshold": np.float64(case["threshold"]), "calls_sample0": calls0, "xpca0": xpca,
       "z0": z, "n0": n.astype(np.int16), "masked_bins": case["masked_bins"],
       "n_calls_all": np.array([len(o["results_calls"]) for o in outs])}
for j, c in enumerate(cfg5_case.GOLDEN_CHROMS):
    out["z_chr%d" % (c + 1)] = zs[c]
    out["r_chr%d" % (c + 1)] = rs[c]
    out["gpu_whole_chr%d" % (c + 1)] = np.float64(whole[j])
    out["gpu_seg_chr%d" % (c + 1)] = np.array([[v, x, y] for v, (x, y) in segs[j]], dtype=np.float64).reshape(-1, 3)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "cfg5_z.npz"), **out)
print("cfg5 z vectors written; calls of sample 0:", calls0[:, :3].tolist(), "lengths",
      [len(zs[c]) for c in cfg5_case.GOLDEN_CHROMS], "%.1f s" % (time.time() - t0))
