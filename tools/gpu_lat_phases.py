"""Phase stamps of the latency-mode kernels for every region of one sample (development aid)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from wisecondor_amd import _lib, distributed
from wisecondor_amd import wisetools as wt
from wisecondor_amd.wisecondor import zThreshold
inp = bench.build_inputs(250000, 100, 1)
bins = inp["masked_bins"]
idx, dst = wt.getReference(inp["corrected"], bins, np.cumsum(bins), 100, 1, 1)
ref = wt.Reference(idx, dst, inp["chrom_bins"], bins, inp["mask"], inp["pca_mean"], inp["pca_components"], binsize=250000)
thr = float(zThreshold([int(v) for v in bins], 1000, None))
counts = wt.samples_to_counts(inp["tests"], inp["chrom_bins"])
tb = distributed.TestBatch(ref, torch.from_numpy(counts[:1].copy()).cuda(), thr, max_calls=256)
os.environ["WC_TEST_LATENCY_MODE"] = "2"
lib, ctx = _lib.load(), _lib.context(0)
for _ in range(3):
    tb.run()
out = np.zeros(64, dtype=np.uint64)
for region in range(22):
    _lib.check(lib.wc_debug_times(ctx, region + 1, None))
    tb.run()
    _lib.check(lib.wc_debug_times(ctx, 0, _lib.ptr(out)))
    t = out.astype(np.int64)
    d = lambda a, b: (t[b] - t[a]) / 2100.0 if t[a] and t[b] else float("nan")      # clock64: shader cycles, ~2.1 GHz
    print("region %2d setup: clean %.1f prefix %.1f whole %.1f | tree: load %.1f collect %.1f decide %.1f rest->calls %.1f calls %.1f | child collect %.1f decide %.1f (us)"
          % (region, d(0, 1), d(1, 2), d(2, 3), d(8, 9), d(9, 10), d(10, 11), d(11, 16), d(16, 17), d(13, 14), d(14, 15)))
    out[:] = 0
print("calls", tb.calls[0, :int(tb.n_calls[0])].cpu().numpy()[:, :3].tolist())
