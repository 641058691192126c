"""Time the newref stages separately (events), optionally with WC_DEBUG_FINISH phase skipping."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from wisecondor_amd import _lib, synth, distributed
lib = _lib.load(); ctx = _lib.context(0)
for binsize, S in [(250000, 100), (50000, 600)]:
    data, bins, sums = synth.corrected_matrix(binsize, S, seed=0)
    X = torch.from_numpy(data).cuda()
    for order in (0, 1):
        job = distributed.NewrefJob(ctx, X, bins, 100, order)
        st = job.st
        for dbg in ("0", "1", "2", "4", "7"):
            os.environ["WC_DEBUG_FINISH"] = dbg
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            for it in range(2):
                ev[0].record(); st.prepare(); ev[1].record(); st.thresholds(0, st.n_bins); ev[2].record()
                st.collect(0, st.n_bins, 0, 1); ev[3].record(); st.finish(0, st.n_bins, job.idx, job.dst); ev[4].record()
                torch.cuda.synchronize()
            print("S", S, "order", order, "dbg", dbg, "prep %.3f thr %.3f collect %.3f finish %.3f ms" % tuple(ev[i].elapsed_time(ev[i+1]) for i in range(4)), flush=True)
