#!/usr/bin/env python3
"""End-to-end files/s of `testbatch`: N converted-sample files on local disk ->
GPU batches -> N result files, with the decode / encode pools at 1 thread (the reference's serial
np.load loop, wisecondor.py:193-196) and at several pool sizes.  Also times the pooled sample load
of `newrefprep`.  Run on the GPU box; prints one JSON line."""
import argparse
import json
import os
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

ap = argparse.ArgumentParser()
ap.add_argument("--files", type=int, default=8192)
ap.add_argument("--binsize", type=int, default=250000)
ap.add_argument("--batch", type=int, default=512)
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="wc_ingest_")
profile = synth.bin_profile(a.binsize)
refs = [synth.make_sample(profile, seed=i) for i in range(40)]
_, chrom_bins, mask, corrected, comps, mean, masked_bins = wt.prepReference(refs)
masked_bins = np.asarray(masked_bins, dtype=np.int64)
idx, dst = wt.getReference(corrected, masked_bins, np.cumsum(masked_bins), 100, 1, 1)
refpath = os.path.join(tmp, "reference.npz")
np.savez_compressed(refpath, arguments={}, runtime={}, binsize=float(a.binsize), indexes=idx, distances=dst,
                    chromosome_sizes=np.asarray(chrom_bins), mask=mask, masked_sizes=masked_bins,
                    pca_components=comps, pca_mean=mean)
paths = []
for i in range(a.files):
    p = os.path.join(tmp, "s_%05d.npz" % i)
    if i < 64:
        np.savez_compressed(p, arguments={"binsize": float(a.binsize)}, runtime={}, sample=synth.make_sample(profile, seed=3000 + i), quality={})
    else:
        shutil.copyfile(paths[i % 64], p)       # (the decode cost does not depend on the counts)
    paths.append(p)
out = {"files": a.files, "binsize": a.binsize, "batch": a.batch, "testbatch": {}}
import contextlib
import io
for label, threads, extra_argv, env in (("native_io_16_warmup", 16, [], {}), ("native_io_1", 1, [], {}), ("native_io_4", 4, [], {}), ("native_io_16", 16, [], {}),
                                        ("native_io_16_stored", 16, ["-ziplevel", "0"], {}),
                                        ("native_io_16_level6", 16, ["-ziplevel", "6"], {})):
    outdir = os.path.join(tmp, "out_%s" % label)
    argv = ["testbatch"] + paths + [outdir, refpath, "-batch", str(a.batch), "-io", str(threads)] + extra_argv
    os.environ.update(env)
    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        cli.main(argv)
    wall = time.time() - t0
    for k in env:
        del os.environ[k]
    line = [ln for ln in buf.getvalue().splitlines() if ln.startswith("rank 0")][-1]
    size = sum(os.path.getsize(os.path.join(outdir, f)) for f in os.listdir(outdir)) / max(1, len(os.listdir(outdir)))
    out["testbatch"][label] = {"wall_s_incl_reference_load": wall, "report": line, "bytes_per_result_file": size}
    shutil.rmtree(outdir, ignore_errors=True)
for threads in (1, 16):
    t0 = time.time()
    ingest.load_samples(paths[:256], None, threads=threads)   # (newref's sample load: still the np.load pool)
    out["load_samples_256_files_threads_%d_s" % threads] = time.time() - t0
shutil.rmtree(tmp, ignore_errors=True)
print(json.dumps(out))
