"""Phase clocks of k_seg_walk over one batch (WC_LIB_PATH=wisecondor_amd/ab/lib_walkclk.so, tools/walk_clocks_variant.py):
    python tools/gpu_walk_clocks.py [samples] [binsize]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from wisecondor_amd import _lib, distributed, wisetools as wt
from wisecondor_amd.wisecondor import zThreshold
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 125
binsize = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
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
lib, ctx = _lib.load(), _lib.context(0)
for _ in range(3):
    tb.run()
a = np.zeros(64, dtype=np.uint64); b = np.zeros(64, dtype=np.uint64)
_lib.check(lib.wc_debug_times(ctx, 0, _lib.ptr(a)))
tb.run()
torch.cuda.synchronize()
_lib.check(lib.wc_debug_times(ctx, 0, _lib.ptr(b)))
d = (b - a).astype(np.float64)
wgs, ranges = d[16], d[17]
print("region walks %d, ranges searched %d (%.2f per region)" % (wgs, ranges, ranges / max(wgs, 1)))
names = {1: "tables + seed", 2: "search: near sweep", 3: "search: cell sweeps", 4: "search: rows + windows", 7: "extremes (block reduce)",
         10: "2nd pass: near sweep", 11: "2nd pass: cell sweeps", 12: "2nd pass: rows + windows", 18: "exact candidates", 19: "decide + loop tail"}
tot = sum(d[k] for k in names)
for k, n in names.items():
    print("%-26s %10.0f kticks %5.1f %%  %8.0f ticks per region walk" % (n, d[k] / 1e3, 100 * d[k] / tot, d[k] / max(wgs, 1)))
print("total %.0f ticks per region walk; loud cells per range: 128 x 128 %.1f, 32 x 32 band %.1f" % (tot / max(wgs, 1), d[5] / max(ranges, 1), d[6] / max(ranges, 1)))
