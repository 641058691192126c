#!/usr/bin/env python3
"""tests/golden/cfg5_whole.npz: WHOLE 50 kb samples through the real reference's toolTest
(wisecondor.py:174-281) and through the CPU oracle (dev container only; /root/reference needed).

Inputs: gpurun_out/cfg5_ref.npz (the 50 kb reference set of tools/cfg5_case.py, written on the GPU
box by tools/gpu_cfg5_ref.py -- a `newref` output, i.e. an INPUT of `test`) and the deterministic
test cohort of tools/cfg5_case.py.  Nothing the HIP `test` path computes goes into the fixture.

* REF_SAMPLES run the real reference end to end (fillTri: one np.sum per window, 90 M windows per
  sample -- tens of minutes each): results_calls, results_cwz, threshold_z, asdef and, of
  results_z / results_r, a SHA-256 plus every 97th value.
* ORACLE_SAMPLES run oracle.wc_oracle.test_sample: results_calls only (the oracle is pinned on the
  reference by tests/test_oracle_vs_golden.py, which also compares it with the REF_SAMPLES here).

Each sample is its own process (`--jobs N`); partial results are cached in gpurun_out/cfg5_whole/
so an interrupted run resumes.
"""
import argparse
import contextlib
import hashlib
import io
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cfg5_case  # noqa: E402
from wisecondor_amd import synth  # noqa: E402

REF_SAMPLES = (0, 5)
ORACLE_SAMPLES = (0, 1, 5, 10, 15, 20, 25, 30, 35, 124)
CACHE = os.path.join(ROOT, "gpurun_out", "cfg5_whole")
STRIDE = 97


def reference_arrays():
    z = np.load(os.path.join(ROOT, "gpurun_out", "cfg5_ref.npz"))
    keys = ("binsize", "indexes", "distances", "chromosome_sizes", "mask", "masked_sizes", "pca_mean",
            "pca_components")
    return {k: z[k] for k in keys}, z


def run_reference(i):
    import ref_loader
    wt, wc, _ = ref_loader.load(full_svd=True)
    ref, _ = reference_arrays()
    sample = cfg5_case.test_samples(synth, synth.bin_profile(cfg5_case.BINSIZE), i + 1)[i]
    tmp = tempfile.mkdtemp(prefix="wc_cfg5w_")
    refpath = os.path.join(tmp, "reference.npz")
    np.savez(refpath, **ref)
    sp = os.path.join(tmp, "sample.npz")
    np.savez(sp, sample=sample, quality={}, arguments={"binsize": float(cfg5_case.BINSIZE)}, runtime={})
    op = os.path.join(tmp, "out.npz")
    args = argparse.Namespace(infile=sp, outfile=op, reference=refpath, minzscore=None,
                              chromosomes=list(range(1, 23)), mineffectsize=0, multitest=1000,
                              minrefbins=25, repeats=5)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        try:
            wc.toolTest(args)
        except SystemExit:
            pass
    tz = np.load(op, allow_pickle=True)
    rz = np.concatenate(list(tz["results_z"])).astype(np.float64)
    rr = np.concatenate(list(tz["results_r"])).astype(np.float64)
    out = dict(results_calls=np.asarray(tz["results_calls"], dtype=np.float64).reshape(-1, 5),
               results_cwz=np.asarray(tz["results_cwz"], dtype=np.float64),
               threshold_z=np.float64(tz["threshold_z"]), asdef=np.float64(tz["asdef"]),
               aasdef=np.float64(tz["aasdef"]),
               results_z_sha256=np.array(hashlib.sha256(rz.tobytes()).hexdigest()),
               results_r_sha256=np.array(hashlib.sha256(rr.tobytes()).hexdigest()),
               results_z_sampled=rz[::STRIDE].copy(), results_r_sampled=rr[::STRIDE].copy(),
               results_z_nonzero=np.int64(np.count_nonzero(rz)), seconds=np.float64(time.time() - t0))
    return out


def run_oracle(i):
    from oracle import wc_oracle as wo
    ref, _ = reference_arrays()
    sample = cfg5_case.test_samples(synth, synth.bin_profile(cfg5_case.BINSIZE), i + 1)[i]
    t0 = time.time()
    with np.errstate(all="ignore"):
        res = wo.test_sample(sample, cfg5_case.BINSIZE, ref)
    return dict(results_calls=np.asarray(res["results_calls"], dtype=np.float64).reshape(-1, 5),
                results_cwz=np.asarray(res["results_cwz"], dtype=np.float64),
                asdef=np.float64(res["asdef"]), seconds=np.float64(time.time() - t0))


def work(task):
    kind, i = task
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "%s_%03d.npz" % (kind, i))
    if not os.path.exists(path):
        out = run_reference(i) if kind == "ref" else run_oracle(i)
        np.savez(path + ".tmp.npz", **out)
        os.replace(path + ".tmp.npz", path)
        print(kind, i, "done: %d call(s), %.0f s" % (len(out["results_calls"]), float(out["seconds"])), flush=True)
    return task


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=6)
    ap.add_argument("--collect-only", action="store_true")
    args = ap.parse_args()
    tasks = [("ref", i) for i in REF_SAMPLES] + [("oracle", i) for i in ORACLE_SAMPLES]
    if not args.collect_only:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(args.jobs) as pool:
            for _ in pool.imap_unordered(work, tasks, chunksize=1):
                pass
    ref, z = reference_arrays()
    # cutoff / threshold as the reference derives them from the reference FILE (getOptimalCutoff,
    # wisetools.py:328-336, and wisecondor.py:203-204), computed here by the CPU oracle -- not the values the GPU
    # box happened to store beside the arrays
    from oracle import wc_oracle as wo
    cutoff = wo.get_optimal_cutoff(ref["distances"], 3)[0]
    threshold = wo.z_threshold([int(v) for v in ref["masked_sizes"]], 1000, None)
    out = dict(threshold=np.float64(threshold), cutoff=np.float64(cutoff),
               masked_sizes=np.asarray(z["masked_sizes"], dtype=np.int64),
               distances_sha256=z["distances_sha256"], stride=np.int64(STRIDE),
               ref_samples=np.array(REF_SAMPLES, dtype=np.int64),
               oracle_samples=np.array(ORACLE_SAMPLES, dtype=np.int64))
    for kind, i in tasks:
        part = np.load(os.path.join(CACHE, "%s_%03d.npz" % (kind, i)))
        for k in part.files:
            if k != "seconds":
                out["%s%d_%s" % (kind, i, k)] = part[k]
        out["%s%d_seconds" % (kind, i)] = part["seconds"]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cfg5_whole.npz"), **out)
    print("wrote tests/golden/cfg5_whole.npz")


if __name__ == "__main__":
    main()
