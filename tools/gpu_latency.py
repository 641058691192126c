#!/usr/bin/env python3
"""Single-sample `test` latency (BASELINE config 3) on the GPU box: general path vs latency mode
(hipGraph replay, device-side segmentation rounds), outputs compared bit for bit."""
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
n_ref = 100 if binsize >= 250000 else 40
inp = bench.build_inputs(binsize, n_ref, 6)
bins = inp["masked_bins"]
idx, dst = wt.getReference(inp["corrected"], bins, np.cumsum(bins), 100, 1, 1)
ref = wt.Reference(idx, dst, inp["chrom_bins"], bins, inp["mask"], inp["pca_mean"], inp["pca_components"], binsize=binsize)
thr = float(zThreshold([int(v) for v in bins], 1000, None))
counts = wt.samples_to_counts(inp["tests"], inp["chrom_bins"])
dev = torch.device("cuda", 0)
results = {}
for mode in ("0", "2", "1"):
    os.environ["WC_TEST_LATENCY_MODE"] = mode
    outs = []
    side = torch.cuda.Stream()           # a real stream: the call is captured / replayed on it directly
    for i in range(counts.shape[0]):
        tb = distributed.TestBatch(ref, torch.from_numpy(counts[i:i + 1].copy()).to(dev), thr, max_calls=256)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(3):
                tb.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 30
            for _ in range(n):
                tb.run()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / n
        nc = int(tb.n_calls[0].item())
        outs.append((ms, tb.results_z.cpu().numpy().copy(), tb.results_r.cpu().numpy().copy(), tb.cwz.cpu().numpy().copy(),
                     tb.calls[0, :nc].cpu().numpy().copy(), float(tb.asdef[0].item())))
    results[mode] = outs
    print("latency mode %s: ms per sample %s" % (mode, " ".join("%.4f" % o[0] for o in outs)), flush=True)
for a, b in list(zip(results["0"], results["1"])) + list(zip(results["0"], results["2"])):
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(np.asarray(x).view(np.uint64) if np.asarray(x).dtype == np.float64 else x,
                              np.asarray(y).view(np.uint64) if np.asarray(y).dtype == np.float64 else y)
print("outputs identical; calls per sample", [len(o[4]) for o in results["1"]])
