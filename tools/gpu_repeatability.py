"""Does a batched test give the same bits on every call of a fresh process?  (first call = fresh allocations)
    python tools/gpu_repeatability.py <samples> <binsize> [calls]      ["WC_X=.." env through the shell]"""
import sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench
from wisecondor_amd import _lib, distributed, wisetools as wt
from wisecondor_amd.wisecondor import zThreshold
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 125
binsize = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
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
first = None
import os
for it in range(reps):
    if it == 0 and os.environ.get("TRUTH_ENV"):
        k_, v_ = os.environ["TRUTH_ENV"].split("=")
        os.environ[k_] = v_
    tb.run(); torch.cuda.synchronize()
    if it == 0 and os.environ.get("TRUTH_ENV"):
        del os.environ[os.environ["TRUTH_ENV"].split("=")[0]]
    got = {k: getattr(tb, k).cpu().numpy().copy() for k in ("results_z", "results_r", "cwz", "calls", "n_calls", "asdef")}
    if first is None:
        first = got
        print("call 0: calls %d" % int(got["n_calls"].sum()))
        continue
    line = []
    for k, v in got.items():
        a, b = first[k], v
        if a.dtype.kind == "f":
            same = (a.view(np.int64) == b.view(np.int64)) | (np.isnan(a) & np.isnan(b))
        else:
            same = a == b
        bad = np.argwhere(~same)
        line.append("%s %d differ%s" % (k, len(bad), (" first at %s: %r vs %r" % (bad[0].tolist(), a[tuple(bad[0])], b[tuple(bad[0])])) if len(bad) else ""))
    print("call %d vs call 0: %s" % (it, "; ".join(line)))
    bad = np.argwhere(first["calls"].view(np.int64) != got["calls"].view(np.int64))
    for s_, c_ in sorted(set((int(a), int(b)) for a, b, _ in bad))[:6]:
        print("   sample %d call %d: first %s now %s" % (s_, c_, first["calls"][s_, c_].tolist(), got["calls"][s_, c_].tolist()))
