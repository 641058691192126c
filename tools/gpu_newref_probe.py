"""Quick GPU probe: time the device-resident newref path on the BASELINE configs."""
import sys
import time
import ctypes
import numpy as np
import torch

sys.path.insert(0, ".")
from wisecondor_amd import _lib, synth

lib = _lib.load()
ctx = _lib.context(0)
for binsize, S in [(1000000, 16), (250000, 100), (50000, 600)]:
    data, bins, sums = synth.corrected_matrix(binsize, S, seed=0)
    B = data.shape[0]
    X = torch.from_numpy(data).cuda()
    idx = torch.empty((B, 100), dtype=torch.int32, device="cuda")
    dst = torch.empty((B, 100), dtype=torch.float64, device="cuda")
    bins_c = np.ascontiguousarray(bins, dtype=np.int64)
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        _lib.check(lib.wc_get_reference_dev(ctx, None, X.data_ptr(), B, S, _lib.ptr(bins_c), 22, 100, 0, 0, B,
                                            idx.data_ptr(), dst.data_ptr()))
        torch.cuda.synchronize()
        dt = time.time() - t0
    out = np.zeros(8, dtype=np.int64)
    lib.wc_newref_stats(ctx, _lib.ptr(out))
    pairs = float(B) * B - float((bins.astype(np.float64) ** 2).sum())
    print("binsize", binsize, "S", S, "B", B, "time %.4f s" % dt, "pairs/s %.3e" % (pairs / dt),
          "TF(sym) %.2f" % (pairs * S / dt / 1e12), "stats", out.tolist(), flush=True)
