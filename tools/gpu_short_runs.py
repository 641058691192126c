"""Why a 20-pass timing of the newref pass reads higher than a 200-pass one: python3 tools/gpu_short_runs.py
(the driver runs bench.py with --steps 20 --warmup 5).  Prints the per-pass time of consecutive groups of passes,
each group between two synchronizes, after 5 warm-up passes on an idle GPU."""
import sys, time
import torch
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
from wisecondor_amd import _lib, synth, distributed
data, bins, sums = synth.corrected_matrix(250000, 100, seed=0)
X = torch.from_numpy(data).cuda()
job = distributed.NewrefJob(_lib.context(0), X, bins, 100, _lib.SUM_SEQUENTIAL)
job.run(); torch.cuda.synchronize()
for idle in (0.0, 2.0):
    time.sleep(idle)                                   # an idle GPU drops its clocks
    for _ in range(5):
        job.run()
    torch.cuda.synchronize()
    out = []
    for group in (20, 20, 20, 20, 200, 20):
        t0 = time.perf_counter()
        for _ in range(group):
            job.run()
        torch.cuda.synchronize()
        out.append("%d: %.4f" % (group, (time.perf_counter() - t0) / group * 1e3))
    print("after %.0f s idle + 5 warm-up passes, ms per pass by group -- %s" % (idle, ", ".join(out)), flush=True)
