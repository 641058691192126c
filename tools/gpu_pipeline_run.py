"""N batches of the batched test through distributed.TestPipeline at one depth (for kernel traces):
   python3 tools/gpu_pipeline_run.py <depth> <batches> [samples] [binsize]"""
import sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench
from wisecondor_amd import _lib, distributed, wisetools as wt
from wisecondor_amd.wisecondor import zThreshold
depth, nb = int(sys.argv[1]), int(sys.argv[2])
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 128
binsize = int(sys.argv[4]) if len(sys.argv) > 4 else 250000
inp = bench.build_inputs(binsize, 100, ns)
corrected = inp["corrected"]; bins = np.ascontiguousarray(inp["masked_bins"])
X = torch.from_numpy(np.ascontiguousarray(corrected)).cuda()
job = distributed.NewrefJob(_lib.context(0), X, bins, 100, wt.sum_order_of(corrected))
idx, dst = job.run(); torch.cuda.synchronize()
thr = float(zThreshold([int(v) for v in inp["masked_bins"]], 1000, None))
counts = torch.from_numpy(wt.samples_to_counts(inp["tests"], inp["chrom_bins"])).cuda()
ref = wt.Reference(idx.cpu().numpy(), dst.cpu().numpy(), inp["chrom_bins"], inp["masked_bins"], inp["mask"],
                   inp["pca_mean"], inp["pca_components"], binsize=binsize)
with distributed.TestPipeline(ref, thr, depth=depth, max_calls=256) as pipe:
    pipe.run([counts] * (2 * depth))
    torch.cuda.synchronize()
    pipe.run([counts] * nb)
    torch.cuda.synchronize()
