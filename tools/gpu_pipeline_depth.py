"""Throughput of distributed.TestPipeline at several depths: python3 tools/gpu_pipeline_depth.py 128 250000 [batches]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench
from wisecondor_amd import _lib, distributed, wisetools as wt
from wisecondor_amd.wisecondor import zThreshold
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 128
binsize = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 24
inp = bench.build_inputs(binsize, 100, ns)
corrected = inp["corrected"]; bins = np.ascontiguousarray(inp["masked_bins"])
X = torch.from_numpy(np.ascontiguousarray(corrected)).cuda()
job = distributed.NewrefJob(_lib.context(0), X, bins, 100, wt.sum_order_of(corrected))
idx, dst = job.run(); torch.cuda.synchronize()
thr = float(zThreshold([int(v) for v in inp["masked_bins"]], 1000, None))
counts = torch.from_numpy(wt.samples_to_counts(inp["tests"], inp["chrom_bins"])).cuda()
ref = wt.Reference(idx.cpu().numpy(), dst.cpu().numpy(), inp["chrom_bins"], inp["masked_bins"], inp["mask"],
                   inp["pca_mean"], inp["pca_components"], binsize=binsize)
for depth in (2, 4, 6, 8):
    pipe = distributed.TestPipeline(ref, thr, depth=depth, max_calls=256)
    batches = [counts] * nb
    pipe.run(batches[:2 * depth])
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        pipe.run(batches)
        torch.cuda.synchronize(); best = min(best, (time.time() - t0) / nb)
    pipe.close()
    print("%d x %d kb, depth %d: %.3f ms per batch -> %.0f samples/s" % (ns, binsize // 1000, depth, best * 1e3, ns / best), flush=True)
