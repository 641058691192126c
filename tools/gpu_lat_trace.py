"""Runs one sample through the test path N times (for rocprofv3 kernel traces of latency mode)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wisecondor_amd import distributed  # noqa: E402
from wisecondor_amd import wisetools as wt  # noqa: E402
from wisecondor_amd.wisecondor import zThreshold  # noqa: E402
inp = bench.build_inputs(250000, 100, 1)
bins = inp["masked_bins"]
idx, dst = wt.getReference(inp["corrected"], bins, np.cumsum(bins), 100, 1, 1)
ref = wt.Reference(idx, dst, inp["chrom_bins"], bins, inp["mask"], inp["pca_mean"], inp["pca_components"], binsize=250000)
thr = float(zThreshold([int(v) for v in bins], 1000, None))
counts = wt.samples_to_counts(inp["tests"], inp["chrom_bins"])
tb = distributed.TestBatch(ref, torch.from_numpy(counts[:1].copy()).cuda(), thr, max_calls=256)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    tb.run()
torch.cuda.synchronize()
