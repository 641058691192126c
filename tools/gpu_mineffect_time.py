"""Time of `test -mineffectsize` (fillTriMin's median filter, wisetools.py:479-487) on the GPU: one 250 kb
sample and a batch of 16, against the same calls without the filter.  python3 tools/gpu_mineffect_time.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wisecondor_amd import distributed  # noqa: E402
from wisecondor_amd import wisetools as wt  # noqa: E402
from wisecondor_amd.wisecondor import zThreshold  # noqa: E402

binsize = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 16
inp = bench.build_inputs(binsize, 100 if binsize >= 250000 else 40, nmax)
bins = inp["masked_bins"]
idx, dst = wt.getReference(inp["corrected"], bins, np.cumsum(bins), 100, 1, 1)
ref = wt.Reference(idx, dst, inp["chrom_bins"], bins, inp["mask"], inp["pca_mean"], inp["pca_components"], binsize=binsize)
thr = float(zThreshold([int(v) for v in bins], 1000, None))
counts = wt.samples_to_counts(inp["tests"], inp["chrom_bins"])
for ns in (1, nmax):
    for eff in (0.0, 0.01):
        tb = distributed.TestBatch(ref, torch.from_numpy(counts[:ns].copy()).cuda(), thr, max_calls=256, mineffectsize=eff)
        for _ in range(3):
            tb.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            tb.run()
        torch.cuda.synchronize()
        print("%2d sample(s), mineffectsize %.2f: %.3f ms per call, %d calls" %
              (ns, eff, 1e3 * (time.perf_counter() - t0) / 5, int(tb.n_calls.sum())), flush=True)
