"""Time of the collect stage alone (k_gram_glds): python3 tools/gpu_collect_time.py cfg4 ["ENV=val ..." ...]
each variant: prepare + thresholds once, then the tile launch repeated between two events (the lists are reset by a
prepare + thresholds before every timed group, so the appends are the real ones)."""
import os
import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
from wisecondor_amd import _lib, synth, distributed

WL = {"cfg2": (250000, 100), "cfg4": (50000, 600)}
name = sys.argv[1]
variants = sys.argv[2:] or [""]
binsize, n_samples = WL[name]
data, bins, sums = synth.corrected_matrix(binsize, n_samples, seed=0)
X = torch.from_numpy(data).cuda()
B = data.shape[0]
ctx = _lib.context(0)
for var in variants:
    settings = dict(kv.split("=", 1) for kv in var.split()) if var else {}
    os.environ.update(settings)
    job = distributed.NewrefJob(ctx, X, bins, 100, _lib.SUM_SEQUENTIAL)
    st = job.st
    times = []
    for rep in range(8):
        st.prepare()
        st.thresholds(0, B)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        st.collect(0, B, 0, 1)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    times.sort()
    print("%s [%s] collect: min %.4f ms, median %.4f ms" % (name, var or "default", times[0], times[len(times) // 2]), flush=True)
    for k in settings:
        del os.environ[k]
