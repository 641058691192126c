"""Wall time of the `newref` sub-command end to end (files in -> reference file out) and where it goes:
    python tools/gpu_cli_newref_time.py [binsize] [samples]"""
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
from wisecondor_amd import synth  # noqa: E402
from wisecondor_amd import wisecondor as cli  # noqa: E402

binsize = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tmp = tempfile.mkdtemp(prefix="wc_cli_")
profile = synth.bin_profile(binsize)
paths = []
for i in range(n):
    p = os.path.join(tmp, "ref_%03d.npz" % i)
    np.savez_compressed(p, arguments={"binsize": float(binsize)}, runtime={}, sample=synth.make_sample(profile, seed=i), quality={})
    paths.append(p)
for rep in range(2):
    out = os.path.join(tmp, "reference_%d.npz" % rep)
    pr = cProfile.Profile()
    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        if rep:
            pr.enable()
        cli.main(["newref"] + paths + [out])
        if rep:
            pr.disable()
    print("newref %d samples x %d kb: %.2f s wall (run %d), reference file %.1f MB" % (
        n, binsize // 1000, time.time() - t0, rep, os.path.getsize(out) / 1e6), flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
shutil.rmtree(tmp, ignore_errors=True)
