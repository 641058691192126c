"""What one rank of an N-rank newref job does, timed on one GPU (no collectives).

    python tools/gpu_band_times.py [cfg2|cfg4]

rows mode : prepare, thresholds/collect/finish of the rank's row band against all columns
tiles mode: prepare, thresholds of the band, collect of the rank's tile share, finish of the band
            (lists of foreign rows are simply left behind: the exchange is not timed here)
Prints device time per step (events) and host enqueue time per step.
"""
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
from wisecondor_amd import _lib, synth, distributed  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
binsize, n_samples = {"cfg2": (250000, 100), "cfg4": (50000, 600)}[which]
ctx = _lib.context(0)
data, bins, sums = synth.corrected_matrix(binsize, n_samples, seed=0)
X = torch.from_numpy(data).cuda()
job = distributed.NewrefJob(ctx, X, bins, 100, _lib.SUM_SEQUENTIAL)
st = job.st
B = st.n_bins
steps = 20 if which == "cfg2" else 3
for world in (1, 2, 4, 8):
    rb, re = distributed.row_range(world - 1, world, B)
    idx = torch.empty((re - rb, 100), dtype=torch.int32, device="cuda")
    dst = torch.empty((re - rb, 100), dtype=torch.float64, device="cuda")
    for mode in ("rows", "tiles"):
        def step():
            st.prepare()
            st.thresholds(rb, re)
            if mode == "rows":
                st.collect(rb, re, 0, 1)
            else:
                st.collect(0, B, world - 1, world)
            st.finish(rb, re, idx, dst)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("%s world %d %-5s: device %.3f ms/step, host enqueue %.3f ms/step"
              % (which, world, mode, e0.elapsed_time(e1) / steps, 1e3 * t_host / steps), flush=True)
