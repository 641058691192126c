"""Cost of the exact path (python3 tools/gpu_fallback_cost.py [n_outliers]): 600 x 50 kb with n outlier rows that lose
their certificate (the pass with them minus the clean pass), every row of a band through wc_newref_exact_dev in both
summation orders, and refsize 300 (above 256 every row takes the exact path) at 100 samples x 250 kb."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
from wisecondor_amd import _lib, synth, distributed
from wisecondor_amd import wisetools as wt

n_out = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def timed(fn, reps=3):
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return min(out)


data, bins, sums = synth.corrected_matrix(50000, 600, seed=0)
B = data.shape[0]
ctx = _lib.context(0)
X = torch.from_numpy(data).cuda()
job = distributed.NewrefJob(ctx, X, bins, 100, _lib.SUM_SEQUENTIAL)
job.run()
clean = timed(job.run)
print("600 x 50 kb (%d bins), clean pass: %.3f ms, stats %s" % (B, clean, wt.newref_stats(0)), flush=True)
rng = np.random.RandomState(1)
rows = rng.permutation(B)[:n_out]
spoiled = data.copy()
spoiled[rows] *= 25.0               # far from everything: ~all candidates within the slack of the k-th -> no certificate
X2 = torch.from_numpy(spoiled).cuda()
job2 = distributed.NewrefJob(ctx, X2, bins, 100, _lib.SUM_SEQUENTIAL)
job2.run()
with_out = timed(job2.run)
st = wt.newref_stats(0)
print("%d outlier rows: pass %.3f ms -> exact path %.3f ms for %d rows (%.4f ms per row), stats %s"
      % (n_out, with_out, with_out - clean, st["fallback_rows"], (with_out - clean) / max(1, st["fallback_rows"]), st), flush=True)
for order, name in ((_lib.SUM_SEQUENTIAL, "sequential"), (_lib.SUM_PAIRWISE, "pairwise")):
    jb = distributed.NewrefJob(ctx, X, bins, 100, order)
    jb.run()
    idx = torch.empty((4096, 100), dtype=torch.int32, device="cuda")
    dst = torch.empty((4096, 100), dtype=torch.float64, device="cuda")
    jb.st.exact(0, 4096, idx, dst)
    torch.cuda.synchronize()
    ms = timed(lambda: jb.st.exact(0, 4096, idx, dst))
    ops = 4096.0 * B * 600 * 3
    print("wc_newref_exact_dev, 4096 rows x %d candidates x 600 samples, %s order: %.2f ms = %.4f ms per row, %.1f Tops/s "
          "of float64 (sub, mul, add per pair and sample; vector peak 39.3)" % (B, name, ms, ms / 4096, ops / ms / 1e9), flush=True)
d2, b2, _ = synth.corrected_matrix(250000, 100, seed=0)
X3 = torch.from_numpy(d2).cuda()
for k in (100, 300):
    jk = distributed.NewrefJob(ctx, X3, b2, k, _lib.SUM_SEQUENTIAL)
    jk.run()
    ms = timed(jk.run)
    print("100 x 250 kb (%d bins), refsize %d: %.3f ms per pass, stats %s" % (d2.shape[0], k, ms, wt.newref_stats(0)), flush=True)
