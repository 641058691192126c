"""Time of the GPU newref prep (wc_newref_prep_gram + the eigen-solve of the leading pairs, on the GPU (wc_newref_prep_eig)
and by LAPACK on the fetched matrix + wc_newref_prep_finish[_dev]) on random counts.
    python3 tools/gpu_prep_time.py [cfg2|cfg4]"""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "/root/repo")
import torch
from wisecondor_amd import _lib, synth, wisetools as wt
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
binsize, n_s = {"cfg2": (250000, 100), "cfg4": (50000, 600)}[which]
sizes = np.ascontiguousarray(synth.chrom_bins(binsize), dtype=np.int64)
n_total = int(sizes.sum())
rng = np.random.RandomState(0)
prof = rng.gamma(20.0, 1 / 20.0, n_total)
counts = rng.poisson(prof[None, :] * 800.0, size=(n_s, n_total)).astype(np.int32)
lib = _lib.load(); ctx = _lib.context(0)
for it in range(3):
    mask = np.empty(n_total, dtype=np.uint8); mbins = np.empty(len(sizes), dtype=np.int64)
    n_b = ctypes.c_int64(); gram = np.empty((n_s, n_s))
    t0 = time.perf_counter()
    _lib.check(lib.wc_newref_prep_gram(ctx, _lib.ptr(counts), n_s, n_total, _lib.ptr(sizes), len(sizes),
                                       _lib.ptr(mask), _lib.ptr(mbins), ctypes.byref(n_b), _lib.ptr(gram)))
    t1 = time.perf_counter()
    gv, gvec = np.empty(3), np.empty((3, n_s))
    _lib.check(lib.wc_newref_prep_eig(ctx, 3, _lib.ptr(gv), _lib.ptr(gvec)))
    t1b = time.perf_counter()
    t_gpu_eig = t1b - t1
    t1 = t1b
    evals, evecs = wt._leading_eigenpairs(gram, 3)
    t2 = time.perf_counter()
    diff = max(np.abs(np.sign(np.dot(gvec[j], evecs[j])) * gvec[j] - evecs[j]).max() for j in range(3))
    B = n_b.value
    masked = wt._pinned((B, n_s)); corrected_t = wt._pinned((n_s, B)); comps = np.empty((3, B)); mean = np.empty(B)
    _lib.check(lib.wc_newref_prep_finish(ctx, 3, _lib.ptr(evecs), _lib.ptr(evals), _lib.ptr(masked),
                                         _lib.ptr(corrected_t), _lib.ptr(comps), _lib.ptr(mean)))
    t3 = time.perf_counter()
    dm = torch.empty((B, n_s), dtype=torch.float64, device="cuda"); dc = torch.empty_like(dm)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    _lib.check(lib.wc_newref_prep_finish_dev(ctx, 3, _lib.ptr(evecs), _lib.ptr(evals), ctypes.c_void_p(dm.data_ptr()),
                                             ctypes.c_void_p(dc.data_ptr()), _lib.ptr(comps), _lib.ptr(mean)))
    t5 = time.perf_counter()
    print("%s it %d: gram (incl. H2D of the counts, mask round trip) %.1f ms, eigen-solve (leading 3) on the GPU %.2f ms [vectors differ by %.1e] / by LAPACK %.1f ms, finish to pinned host "
          "(D2H of %d MB) %.1f ms, finish device-resident %.1f ms"
          % (which, it, 1e3 * (t1 - t0 - t_gpu_eig), 1e3 * t_gpu_eig, diff, 1e3 * (t2 - t1), (masked.nbytes + corrected_t.nbytes) >> 20, 1e3 * (t3 - t2),
             1e3 * (t5 - t4)), flush=True)
