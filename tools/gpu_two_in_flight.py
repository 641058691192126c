"""Experiment: throughput of the batched test with ONE batch in flight against TWO (two contexts, two streams,
one host thread each: the call reads counts back, i.e. blocks its thread): python3 tools/gpu_two_in_flight.py 128 250000 [batches]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import bench
from wisecondor_amd import _lib, distributed, wisetools as wt
from wisecondor_amd.wisecondor import zThreshold
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 128
binsize = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 16
inp = bench.build_inputs(binsize, 100, ns)
corrected = inp["corrected"]; bins = np.ascontiguousarray(inp["masked_bins"])
X = torch.from_numpy(np.ascontiguousarray(corrected)).cuda()
job = distributed.NewrefJob(_lib.context(0), X, bins, 100, wt.sum_order_of(corrected))
idx, dst = job.run(); torch.cuda.synchronize()
thr = float(zThreshold([int(v) for v in inp["masked_bins"]], 1000, None))
counts = torch.from_numpy(wt.samples_to_counts(inp["tests"], inp["chrom_bins"])).cuda()
lib = _lib.load()
tbs = []
for i in range(2):
    if i:
        _lib._contexts[0] = lib.wc_create(0)          # a second context: its own scratch and side streams
    ref = wt.Reference(idx.cpu().numpy(), dst.cpu().numpy(), inp["chrom_bins"], inp["masked_bins"], inp["mask"],
                       inp["pca_mean"], inp["pca_components"], binsize=binsize)
    tbs.append(distributed.TestBatch(ref, counts, thr, max_calls=256))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for tb, st in zip(tbs, streams):
    with torch.cuda.stream(st):
        tb.run(); tb.run()
torch.cuda.synchronize()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    with torch.cuda.stream(streams[0]):
        for _ in range(nb):
            tbs[0].run()
    torch.cuda.synchronize(); one = (time.time() - t0) / nb
    import threading
    def work(i):
        with torch.cuda.stream(streams[i]):
            for _ in range(nb // 2):
                tbs[i].run()
    t0 = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    t_host = time.time() - t0
    torch.cuda.synchronize(); two = (time.time() - t0) / nb
    print("%d x %d kb: one in flight %.3f ms per batch, two in flight %.3f ms per batch (host enqueue %.3f ms per batch), calls %d / %d"
          % (ns, binsize // 1000, one * 1e3, two * 1e3, t_host / nb * 1e3, int(tbs[0].n_calls.sum()), int(tbs[1].n_calls.sum())), flush=True)
