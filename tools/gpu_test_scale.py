"""Batched test at BASELINE config 5 scale on one GPU: Ns samples x 50 kb (reference built by the GPU newref
on 100 samples to keep the set-up short)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench
from wisecondor_amd import _lib, distributed, wisetools as wt
from wisecondor_amd.wisecondor import zThreshold
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 125
binsize = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
t0 = time.time()
inp = bench.build_inputs(binsize, 100, ns)
print("inputs %.1f s" % (time.time() - t0), flush=True)
corrected = inp["corrected"]; bins = np.ascontiguousarray(inp["masked_bins"])
X = torch.from_numpy(np.ascontiguousarray(corrected)).cuda()
job = distributed.NewrefJob(_lib.context(0), X, bins, 100, wt.sum_order_of(corrected))
idx, dst = job.run(); torch.cuda.synchronize()
ref = wt.Reference(idx.cpu().numpy(), dst.cpu().numpy(), inp["chrom_bins"], inp["masked_bins"], inp["mask"],
                   inp["pca_mean"], inp["pca_components"], binsize=binsize)
thr = float(zThreshold([int(v) for v in inp["masked_bins"]], 1000, None))
counts = torch.from_numpy(wt.samples_to_counts(inp["tests"], inp["chrom_bins"])).cuda()
tb = distributed.TestBatch(ref, counts, thr, max_calls=256)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for it in range(reps):
    torch.cuda.synchronize(); t0 = time.time(); tb.run(); torch.cuda.synchronize(); dt = time.time() - t0
    print("batch of %d samples x %d bins: %.2f ms -> %.0f samples/s, calls %d, mem %.1f GB" % (
        ns, corrected.shape[0], dt * 1e3, ns / dt, int(tb.n_calls.sum()), torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9), flush=True)
