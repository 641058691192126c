"""Phase clocks of k_seg_bound over one 125 x 50 kb batch (needs the variant of tools/phase_bound_variant.py:
WC_LIB_PATH=wisecondor_amd/ab/lib_phase.so python tools/gpu_phase_bound.py)."""
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
_lib.check(lib.wc_debug_times(ctx, 0, _lib.ptr(b)))
d = (b - a).astype(np.float64)
d = d[:16] + d[16:32] + d[32:48] + d[48:64]
names = ["stage rows", "near values", "near blocks", "far level 1", "far level 2", "barrier after sweep", "queue pass",
         "reduce", "stage tables (per wg)"]
chunks, wgs = d[9], d[10]
print("row blocks %d, workgroups %d" % (chunks, wgs))
tot = d[:9].sum()
for i, n in enumerate(names):
    print("%-24s %10.0f kticks  %5.1f %%   %7.0f ticks per %s" % (n, d[i] / 1e3, 100 * d[i] / tot, d[i] / (wgs if i == 8 else chunks),
                                                                 "workgroup" if i == 8 else "row block"))
print("sum per row block %.0f ticks" % (tot / chunks))
