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
before = np.zeros(64 + 8192 + 16384, dtype=np.uint64)
_lib.check(lib.wc_debug_times(ctx, -7, _lib.ptr(before)))      # (stamps of workgroups that leave at once stay from earlier runs)
_lib.check(lib.wc_debug_times(ctx, 0, _lib.ptr(a)))
tb.run()
torch.cuda.synchronize()
_lib.check(lib.wc_debug_times(ctx, 0, _lib.ptr(b)))
d = (b - a).astype(np.float64)
wgs, ranges = d[16], d[17]
print("region walks %d, ranges searched %d (%.2f per region)" % (wgs, ranges, ranges / max(wgs, 1)))
names = {1: "tables + seed", 2: "search: near sweep", 3: "search: cell sweeps", 4: "search: rows + windows", 7: "extremes (block reduce)",
         20: "near: wait + stage", 22: "near: bounds + pushes", 23: "near: queue drain",
         10: "2nd pass: near sweep", 11: "2nd pass: cell sweeps", 12: "2nd pass: rows + windows", 18: "exact candidates", 19: "decide + loop tail"}
tot = sum(d[k] for k in names)
for k, n in names.items():
    print("%-26s %10.0f kticks %5.1f %%  %8.0f ticks per region walk" % (n, d[k] / 1e3, 100 * d[k] / tot, d[k] / max(wgs, 1)))
print("near-sweep trips per range %.2f, queued (row, block) pairs per trip %.1f" % (d[24] / max(ranges, 1), d[25] / max(d[24], 1)))
print("total %.0f ticks per region walk; loud cells per range: 128 x 128 %.1f, 32 x 32 band %.1f" % (tot / max(wgs, 1), d[5] / max(ranges, 1), d[6] / max(ranges, 1)))

# the workgroups' lives (100 MHz wall clock): how full the 1 024 slots are over the kernel's span
big = np.zeros(64 + 8192 + 16384, dtype=np.uint64)
_lib.check(lib.wc_debug_times(ctx, -7, _lib.ptr(big)))
n_wg = 4096
st = big[64:64 + 2 * n_wg:2].astype(np.float64); en = big[65:65 + 2 * n_wg:2].astype(np.float64)
ok = (st > 0) & (en >= st)
ok &= big[64:64 + 2 * n_wg:2] != before[64:64 + 2 * n_wg:2]      # (only the workgroups that ran in THIS call)
st, en = st[ok], en[ok]
t0 = st.min(); span = en.max() - t0
print("workgroups %d: span %.1f us, sum of lives %.0f us (= %.0f slots busy on average), longest life %.1f us, mean %.1f us" % (
    len(st), span / 100, (en - st).sum() / 100, (en - st).sum() / span, (en - st).max() / 100, (en - st).mean() / 100))
for f in range(10):
    t = t0 + span * (f + 0.5) / 10
    print("  at %3d %% of the span: %4d workgroups alive, %4d not started" % (10 * f + 5, int(((st <= t) & (en > t)).sum()), int((st > t).sum())))
samples = ns
for c in range(0, n_wg // samples):
    d = (en - st)[c * samples:(c + 1) * samples] if ok.all() else None
    if d is not None and len(d): print("  chromosome slot %2d: life mean %.1f max %.1f us, starts %.1f .. %.1f us" % (c, d.mean() / 100, d.max() / 100, (st[c * samples:(c + 1) * samples].min() - t0) / 100, (st[c * samples:(c + 1) * samples].max() - t0) / 100))

# the longest-lived workgroups: ranges searched, ticks of the cell sweeps / rows + windows / exact candidates, loud cells
ext = big[64 + 8192:]
life = np.where(ok, big[65:65 + 2 * n_wg:2].astype(np.float64) - big[64:64 + 2 * n_wg:2].astype(np.float64), 0)
for b in np.argsort(-life)[:12]:
    print("  workgroup %4d: start %6.1f life %6.1f us, %3d ranges, kticks: cell sweeps %5.0f, rows + windows %5.0f, exact candidates %5.0f; loud cells 128 x 128 %5d, band %5d" % (
        b, (float(big[64 + 2 * b]) - t0) / 100, life[b] / 100, int(ext[b]), float(ext[4096 + b] >> np.uint64(32)) / 1e3, float(ext[4096 + b] & np.uint64(0xFFFFFFFF)) / 1e3,
        float(ext[8192 + b]) / 1e3, int(ext[12288 + b] >> np.uint64(32)), int(ext[12288 + b] & np.uint64(0xFFFFFFFF))))
