"""Cost of exact-fallback rows: 600 x 50 kb with a few outlier rows (python3 tools/gpu_fallback_cost.py [n_outliers])."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
from wisecondor_amd import _lib, synth, distributed
from wisecondor_amd import wisetools as wt
n_out = int(sys.argv[1]) if len(sys.argv) > 1 else 5
data, bins, sums = synth.corrected_matrix(50000, 600, seed=0)
rng = np.random.RandomState(1)
rows = rng.permutation(data.shape[0])[:n_out]
data[rows] *= 25.0
X = torch.from_numpy(data).cuda()
job = distributed.NewrefJob(_lib.context(0), X, bins, 100, _lib.SUM_SEQUENTIAL)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(3):
    ev[0].record(); idx, dst = job.run(); ev[1].record(); torch.cuda.synchronize()
    print("step %d: %.3f ms, stats %s" % (it, ev[0].elapsed_time(ev[1]), wt.newref_stats(0)), flush=True)
