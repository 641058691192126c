"""Where testbatch's wall time goes: python tools/gpu_ingest_profile.py [files] [batch]"""
import cProfile
import contextlib
import io
import os
import pstats
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wisecondor_amd import ingest, synth  # noqa: E402
from wisecondor_amd import wisecondor as cli  # noqa: E402
from wisecondor_amd import wisetools as wt  # noqa: E402

files = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
binsize = 250000
tmp = tempfile.mkdtemp(prefix="wc_ingest_")
profile = synth.bin_profile(binsize)
refs = [synth.make_sample(profile, seed=i) for i in range(40)]
_, chrom_bins, mask, corrected, comps, mean, masked_bins = wt.prepReference(refs)
masked_bins = np.asarray(masked_bins, dtype=np.int64)
idx, dst = wt.getReference(corrected, masked_bins, np.cumsum(masked_bins), 100, 1, 1)
refpath = os.path.join(tmp, "reference.npz")
np.savez(refpath, arguments={}, runtime={}, binsize=float(binsize), indexes=idx, distances=dst,
         chromosome_sizes=np.asarray(chrom_bins), mask=mask, masked_sizes=masked_bins, pca_components=comps, pca_mean=mean)
base = [synth.make_sample(profile, seed=3000 + i) for i in range(64)]
paths = []
for i in range(files):
    p = os.path.join(tmp, "s_%05d.npz" % i)
    np.savez_compressed(p, arguments={"binsize": float(binsize)}, runtime={}, sample=base[i % 64], quality={})
    paths.append(p)
for rep in range(2):
    outdir = os.path.join(tmp, "out%d" % rep)
    argv = ["testbatch"] + paths + [outdir, refpath, "-batch", str(batch), "-io", "16"]
    buf = io.StringIO()
    pr = cProfile.Profile()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        if rep == 1:
            pr.enable()
        cli.main(argv)
        if rep == 1:
            pr.disable()
    print([ln for ln in buf.getvalue().splitlines() if ln.startswith("rank 0")][-1], "wall %.3f" % (time.time() - t0), flush=True)
    shutil.rmtree(outdir, ignore_errors=True)
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
shutil.rmtree(tmp, ignore_errors=True)
