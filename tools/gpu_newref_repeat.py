"""Does the device-resident newref pass give the same bits on every call?  python tools/gpu_newref_repeat.py cfg2 [calls]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wisecondor_amd import _lib, synth, distributed  # noqa: E402

WL = {"cfg1": (1000000, 16), "cfg2": (250000, 100), "cfg4": (50000, 600)}
name = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 100
binsize, n_samples = WL[name]
data, bins, sums = synth.corrected_matrix(binsize, n_samples, seed=0)
X = torch.from_numpy(data).cuda()
job = distributed.NewrefJob(_lib.context(0), X, bins, 100, _lib.SUM_SEQUENTIAL)
seen = {}
for it in range(calls):
    idx, dst = job.run()
    torch.cuda.synchronize()
    h = hashlib.sha256(idx.cpu().numpy().tobytes() + dst.cpu().numpy().tobytes()).hexdigest()[:12]
    seen[h] = seen.get(h, 0) + 1
print("%s: %d calls, results %s" % (name, calls, seen))
