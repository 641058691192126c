"""One worker process of bench.py's `cpu_baseline` leg (TEST / MEASUREMENT INFRASTRUCTURE ONLY).

    python oracle/cpu_baseline.py newref <dir> <part> <parts> <rows>
    python oracle/cpu_baseline.py test   <dir> <part> <parts> <rows>
    python oracle/cpu_baseline.py segments <dir> <part> <parts> <rows>

Runs the CPU oracle (oracle/wc_oracle.py, the reference's own algorithmic structure) on a
bounded share of the benchmark workload that bench.py left in <dir>, and prints one JSON line
with the seconds it took.  `newref`: worker p of n takes <rows> target rows from the start of
the reference's part p of n (getPart, wisetools.py:358-361) against all candidates -- the
reference's `-cpus n` process model (wisecondor.py:47-56) on a bounded row count.  `test`:
worker p tests sample p (one sample per process; the reference's `test` is single-process).
`segments`: worker p runs the part of `test` that dominates it -- fillTri + segmentTri
(wisetools.py:466-472, triarray.py:59-84), one np.sum per window -- on the cleaned z vector
<dir>/region_<p>.npy of one chromosome (config 5 on one sample x the three
longest chromosomes, scaled by window count).
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wc_oracle as wo  # noqa: E402


def wait_for_go(folder, part):
    """Report ready, then wait for bench.py's common start signal (absent when run by hand)."""
    open(os.path.join(folder, "ready_%d" % part), "w").close()
    if os.environ.get("WC_CPU_BASELINE_NO_WAIT"):
        return
    t_end = time.time() + 600
    while not os.path.exists(os.path.join(folder, "go")) and time.time() < t_end:
        time.sleep(0.005)


def newref(folder, part, parts, rows):
    corrected = np.load(os.path.join(folder, "corrected.npy"))       # Fortran order preserved
    ref = np.load(os.path.join(folder, "reference.npz"))
    bins = [int(v) for v in ref["bins"]]
    sums = [int(v) for v in np.cumsum(bins)]
    k = int(ref["k"])
    B = corrected.shape[0]
    lo, hi = wo.get_part(part, parts, B)
    hi = min(hi, lo + rows)
    wait_for_go(folder, part)
    # the oracle's driver works on whole parts: express [lo, hi) as part 1 of 1 of a row window by
    # calling the per-chromosome kernel exactly as get_reference does (wisetools.py:373-390)
    t0 = time.perf_counter()
    idx_rows = []
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        for chrom, start, end in wo.split_by_chrom(lo, hi, sums):
            start, end = max(start, lo), min(end, hi)
            c_lo, c_hi = sums[chrom] - bins[chrom], sums[chrom]
            chrom_data = np.concatenate((corrected[:c_lo, :], corrected[c_hi:, :]))
            idx, _ = wo.get_ref_for_bins(k, start, end, corrected, chrom_data)
            idx_rows.extend(idx)
    seconds = time.perf_counter() - t0
    if part == 0 and parts == 1:
        np.save(os.path.join(folder, "newref_0.npy"), np.array(idx_rows, dtype=np.int32).reshape(-1, k))
    print(json.dumps({"rows": [int(lo), int(hi)], "seconds": seconds}))


def test(folder, part):
    ref = np.load(os.path.join(folder, "reference.npz"))
    reference = {key: ref[key] for key in ("indexes", "distances", "chromosome_sizes", "mask", "masked_sizes",
                                           "pca_mean", "pca_components")}
    reference["binsize"] = np.float64(ref["binsize"])
    stored = np.load(os.path.join(folder, "sample_%d.npz" % part))
    sample = {key: stored[key] for key in stored.files}
    wait_for_go(folder, part)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        out = wo.test_sample(sample, float(ref["binsize"]), reference)
    seconds = time.perf_counter() - t0
    calls = np.asarray(out["results_calls"], dtype=np.float64).reshape(-1, 5)
    print(json.dumps({"seconds": seconds, "calls": calls.tolist()}))


def segments(folder, part):
    z = np.load(os.path.join(folder, "region_%d.npy" % part))
    thr = float(np.load(os.path.join(folder, "region_thr.npy")))
    wait_for_go(folder, part)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        tri = wo.fill_tri(z)
        segs = wo.segment_tri(tri, len(z), thr, 3)
        whole = float(tri[len(z) - 1])
    seconds = time.perf_counter() - t0
    print(json.dumps({"seconds": seconds, "bins": int(len(z)), "windows": int(len(z)) * (int(len(z)) + 1) // 2,
                      "segments": [[float(v), int(x), int(y)] for v, (x, y) in segs], "whole": whole}))


if __name__ == "__main__":
    what, folder, part, parts, rows = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    if what == "newref":
        newref(folder, part, parts, rows)
    elif what == "segments":
        segments(folder, part)
    else:
        test(folder, part)
