"""Wall-clock step time of the whole device-resident newref pass (what bench.py's `value` times):
    python tools/gpu_newref_step.py cfg2 ["ENV=val ..." ...]
each variant: 30 warm-up passes, then 200 passes between two synchronizes; results must agree."""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wisecondor_amd import _lib, synth, distributed  # noqa: E402

WL = {"cfg1": (1000000, 16), "cfg2": (250000, 100), "cfg4": (50000, 600)}
name = sys.argv[1]
variants = sys.argv[2:] or [""]
binsize, n_samples = WL[name]
data, bins, sums = synth.corrected_matrix(binsize, n_samples, seed=0)
X = torch.from_numpy(data).cuda()
ctx = _lib.context(0)
steps = 200 if name != "cfg4" else 20
seen = set()
side = torch.cuda.Stream() if os.environ.get("WC_STEP_STREAM") else None
if side is not None:
    torch.cuda.set_stream(side)
for var in variants:
    settings = dict(kv.split("=", 1) for kv in var.split()) if var else {}
    os.environ.update(settings)
    job = distributed.NewrefJob(ctx, X, bins, 100, _lib.SUM_SEQUENTIAL)
    for _ in range(30):
        job.run()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            idx, dst = job.run()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    h = hashlib.sha256(idx.cpu().numpy().tobytes() + dst.cpu().numpy().tobytes()).hexdigest()[:12]
    seen.add(h)
    print("%s [%s] %.4f ms per pass, result %s" % (name, var or "default", best * 1e3, h), flush=True)
    for k in settings:
        del os.environ[k]
assert len(seen) == 1, seen
