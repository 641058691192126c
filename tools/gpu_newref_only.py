"""newref only (device resident), for profiling: python3 tools/gpu_newref_only.py cfg4 2 [order]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
from wisecondor_amd import _lib, synth, distributed
wl = {"cfg1": (1000000, 16), "cfg2": (250000, 100), "cfg4": (50000, 600)}[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
order = int(sys.argv[3]) if len(sys.argv) > 3 else 1
data, bins, sums = synth.corrected_matrix(wl[0], wl[1], seed=0)
X = torch.from_numpy(data).cuda()
job = distributed.NewrefJob(_lib.context(0), X, bins, 100, order)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(steps):
    ev[0].record(); job.run(); ev[1].record(); torch.cuda.synchronize()
    print("step %d: %.3f ms" % (it, ev[0].elapsed_time(ev[1])), flush=True)
