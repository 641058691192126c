"""Lone-batch time of the batched test for the library WC_LIB_PATH selects (A/B of two builds: run once per build):
    python tools/gpu_test_ab.py <samples> <binsize> [runs]     -> min / median ms per batch over `runs` calls"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench
from wisecondor_amd import _lib, distributed, wisetools as wt
from wisecondor_amd.wisecondor import zThreshold
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 125
binsize = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 40
inp = bench.build_inputs(binsize, 100, ns)
corrected = inp["corrected"]; bins = np.ascontiguousarray(inp["masked_bins"])
X = torch.from_numpy(np.ascontiguousarray(corrected)).cuda()
job = distributed.NewrefJob(_lib.context(0), X, bins, 100, wt.sum_order_of(corrected))
idx, dst = job.run(); torch.cuda.synchronize()
ref = wt.Reference(idx.cpu().numpy(), dst.cpu().numpy(), inp["chrom_bins"], inp["masked_bins"], inp["mask"],
                   inp["pca_mean"], inp["pca_components"], binsize=binsize)
thr = float(zThreshold([int(v) for v in inp["masked_bins"]], 1000, None))
counts = torch.from_numpy(wt.samples_to_counts(inp["tests"], inp["chrom_bins"])).cuda()
tb = distributed.TestBatch(ref, counts, thr, max_calls=256)
for _ in range(5):
    tb.run()
torch.cuda.synchronize()
ts = []
for _ in range(runs):
    t0 = time.perf_counter(); tb.run(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
import os
print("%s: %d x %d kb: min %.4f median %.4f ms per batch, calls %d" % (os.environ.get("WC_LIB_PATH", "default"), ns, binsize // 1000,
      min(ts), float(np.median(ts)), int(tb.n_calls.sum())), flush=True)
