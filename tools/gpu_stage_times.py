"""Per-stage times of the device-resident newref path (events on the launch stream).

    python tools/gpu_stage_times.py            # cfg2 and cfg4 shapes, both numpy summation orders
"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
from wisecondor_amd import _lib, synth, distributed  # noqa: E402

ctx = _lib.context(0)
for binsize, n_samples in [(250000, 100), (50000, 600)]:
    data, bins, sums = synth.corrected_matrix(binsize, n_samples, seed=0)
    X = torch.from_numpy(data).cuda()
    for order in (_lib.SUM_PAIRWISE, _lib.SUM_SEQUENTIAL):
        job = distributed.NewrefJob(ctx, X, bins, 100, order)
        st = job.st
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        for _ in range(3):
            ev[0].record()
            st.prepare()
            ev[1].record()
            st.thresholds(0, st.n_bins)
            ev[2].record()
            st.collect(0, st.n_bins, 0, 1)
            ev[3].record()
            st.finish(0, st.n_bins, job.idx, job.dst)
            ev[4].record()
            torch.cuda.synchronize()
        t = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
        print("S %d order %d: prepare %.3f thresholds %.3f collect %.3f finish %.3f total %.3f ms"
              % (n_samples, order, t[0], t[1], t[2], t[3], sum(t)), flush=True)
