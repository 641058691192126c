#!/usr/bin/env python3
"""Time the REAL reference against the CPU oracle on identical inputs (development container only:
needs /root/reference) and write the ratios bench.py attaches to its cpu_baseline object
(profiles/<round>_oracle_vs_reference.json).  BASELINE.md section 4(1):

  newref  getReference on a fixed slice of target rows x all candidates (cfg2 prep seam of the
          cfg3 golden: 100 samples x 250 kb, Fortran ordered), work linear in rows
  test    fillTri + segmentTri (90 % of `test`) on the three longest 250 kb chromosomes of one sample
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402
from oracle import wc_oracle as wo  # noqa: E402

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"
wt, wc, tri = ref_loader.load(full_svd=True)
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg3_250kb.npz"))
X = np.asfortranarray(g["prep_correctedData"])
bins = [int(v) for v in g["prep_maskedChromBins"]]
sums = [int(v) for v in np.cumsum(bins)]
parts = 24                                   # part 5 of 24: 462 target rows
res = {"host": "development container, %d cores" % (os.cpu_count() or 0), "numpy": np.__version__}
with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
    t0 = time.perf_counter(); ri, rd = wt.getReference(X, bins, sums, 100, 5, parts); t_ref = time.perf_counter() - t0
    t0 = time.perf_counter(); oi, od = wo.get_reference(X, bins, sums, 100, 5, parts); t_or = time.perf_counter() - t0
assert np.array_equal(np.asarray(ri), oi) and np.array_equal(np.asarray(rd), od)
res["newref"] = {"rows": int(oi.shape[0]), "reference_s": t_ref, "oracle_s": t_or, "oracle_over_reference": t_or / t_ref}
z = g["t_strong5_rep5_z"]
z = z[np.isfinite(z)]
offs = np.concatenate([[0], np.cumsum(bins)])
t_ref = t_or = 0.0
for c in (0, 1, 2):
    zc = np.ascontiguousarray(z[offs[c]:offs[c + 1]][:900])
    with np.errstate(all="ignore"):
        t0 = time.perf_counter(); tr = wt.fillTri(zc); segs = tr.segmentTri(5.0, 3); t_ref += time.perf_counter() - t0
        t0 = time.perf_counter(); to = wo.fill_tri(zc); so = wo.segment_tri(to, len(zc), 5.0, 3); t_or += time.perf_counter() - t0
    assert [(x, y) for _, (x, y) in segs] == [(x, y) for _, (x, y) in so]
res["test_fill_and_segment"] = {"chromosomes": 3, "reference_s": t_ref, "oracle_s": t_or,
                                "oracle_over_reference": t_or / t_ref}
res["note"] = ("time of the oracle (kind 'port') divided by the time of the real reference on the same inputs; "
               "multiply a port throughput by this ratio to estimate the reference's own throughput on that host")
path = os.path.join(ROOT, "profiles", "%s_oracle_vs_reference.json" % ROUND)
json.dump(res, open(path, "w"), indent=1)
print(json.dumps(res, indent=1))
