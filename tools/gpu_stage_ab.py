"""A/B stage times of the device-resident newref pass under environment switches.

    python tools/gpu_stage_ab.py cfg2,cfg4 "" "WC_NEWREF_SHARD=rows" ...

Every variant (a space-separated list of NAME=value settings, "" = defaults) runs the same job;
prints the mean milliseconds between the stage marks and checks that all variants deliver the
same indexes and distances bit for bit."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wisecondor_amd import _lib, synth, distributed  # noqa: E402

WL = {"cfg1": (1000000, 16), "cfg2": (250000, 100), "cfg4": (50000, 600)}
names = sys.argv[1].split(",")
variants = sys.argv[2:] or [""]
ctx = _lib.context(0)
for name in names:
    binsize, n_samples = WL[name]
    data, bins, sums = synth.corrected_matrix(binsize, n_samples, seed=0)
    X = torch.from_numpy(data).cuda()
    steps = 20 if name != "cfg4" else 5
    digests = []
    for order in (_lib.SUM_SEQUENTIAL, _lib.SUM_PAIRWISE):
        for var in variants:
            settings = dict(kv.split("=", 1) for kv in var.split()) if var else {}
            os.environ.update(settings)
            job = distributed.NewrefJob(ctx, X, bins, 100, order)
            for _ in range(3):
                job.run()
            torch.cuda.synchronize()
            marks = []
            for _ in range(steps):
                idx, dst = job.run(timing=True)
                marks.append(job.last_marks)
            torch.cuda.synchronize()
            runs = []
            for m in marks:
                job.last_marks = m
                runs.append(job.stage_ms())
            mean = {k: float(np.mean([r[k] for r in runs])) for k in runs[0]}
            h = hashlib.sha256(idx.cpu().numpy().tobytes() + dst.cpu().numpy().tobytes()).hexdigest()[:12]
            digests.append((order, h))
            stats = {}
            out = np.zeros(8, dtype=np.int64)
            _lib.load().wc_newref_stats(ctx, _lib.ptr(out))
            print("%s order %d [%s] total %.4f ms  %s  result %s fast %d fallback %d rescored %d" % (
                name, order, var or "default", sum(mean.values()),
                " ".join("%s %.4f" % (k.split("->")[1], v) for k, v in mean.items()), h, out[0], out[1], out[4]), flush=True)
            for k in settings:
                del os.environ[k]
    for order in (0, 1):
        hs = {h for o, h in digests if o == order}
        assert len(hs) == 1, "variants disagree for order %d: %s" % (order, digests)
print("all variants agree")
