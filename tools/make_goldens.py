#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the upstream reference (dev container only).

The reference has no tests or fixtures of its own, so
parity is pinned on its behaviour: this script imports the reference through
tools/ref_loader.py (lib2to3 translation in a temp dir + the shims of
SURVEY.md App. C, PCA forced to the deterministic full-SVD solver), feeds it
seeded synthetic inputs and stores inputs + outputs as arrays.  Only data is
written under tests/golden/; no reference source text is.

Run:  python tools/make_goldens.py      (needs /root/reference; ~1 min)
"""
import argparse
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402
from wisecondor_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def obj_array(items):
    out = np.empty(len(items), dtype=object)
    for i, v in enumerate(items):
        out[i] = v
    return out


# ---------------------------------------------------------------- newref ----
def newref_cases(wt):
    """Kernel-level getReference goldens (wisetools.py:364-398)."""
    cases = {}
    rng = np.random.RandomState(42)

    def run(name, data, bins, k, parts_list):
        bins = np.asarray(bins, dtype=np.int64)
        sums = np.cumsum(bins)
        cases[name + "_data"] = data
        cases[name + "_bins"] = bins
        cases[name + "_k"] = np.int64(k)
        cases[name + "_parts"] = np.array(parts_list, dtype=np.int64)
        for parts in parts_list:
            for part in range(1, parts + 1):
                with quiet(), np.errstate(all="ignore"):
                    idx, dst = wt.getReference(data, list(bins), list(sums), k, part, parts)
                idx = np.asarray(idx, dtype=np.int32).reshape(-1, k)
                dst = np.asarray(dst, dtype=np.float64).reshape(-1, k)
                cases["%s_idx_%d_%d" % (name, part, parts)] = idx
                cases["%s_dst_%d_%d" % (name, part, parts)] = dst

    # A: plain tiny genome, several part splits (incl. one that cuts inside chromosomes)
    bins = rng.randint(10, 41, size=22)
    data = 1.0 + 0.02 * rng.standard_normal((int(bins.sum()), 12))
    run("plain", data, bins, 10, [1, 3, 7])

    # B: exact ties -- duplicated rows spread over other chromosomes, 4-way ties
    data_t = data.copy()
    sums = np.cumsum(bins)
    src = 3
    for chrom in (2, 5, 9, 15):
        data_t[sums[chrom] - 2] = data_t[src]
        data_t[sums[chrom] - 5] = data_t[src]
    data_t[sums[20] - 1] = data_t[sums[0] + 1]
    run("ties", data_t, bins, 10, [1, 2])

    # C: fewer candidates than k -> -1 / 1e10 padding
    data_c = 1.0 + 0.05 * rng.standard_normal((12, 6))
    run("fewcand", data_c, [5, 4, 3], 10, [1])

    # D: NaN row, inf row and a far-away row whose distances exceed 1e10
    data_d = data.copy()
    data_d[7, 3] = np.nan
    data_d[sums[4] + 2, 0] = np.inf
    data_d[sums[10] + 1, :] = 2.0e5
    run("special", data_d, bins, 10, [1])

    # E: all rows identical (every distance is exactly 0: pure index order)
    data_e = np.ones((int(bins.sum()), 5)) * 1.25
    run("allsame", data_e, bins, 10, [1])

    # F: S = 600-like summation depth (pairwise split > 128) on a small genome
    bins_f = rng.randint(6, 15, size=22)
    data_f = 1.0 + 0.02 * rng.standard_normal((int(bins_f.sum()), 300))
    run("deep", data_f, bins_f, 20, [1])
    return cases


# -------------------------------------------------- layout-dependent sums ----
def layout_cases(wt):
    """Two places where numpy's summation order follows the memory layout / the length of the run,
    pinned on the reference itself (found by the randomised sweeps of round 1):
      * getReference on Fortran-ordered data when the rows before and after a chromosome are
        single-row pieces (np.concatenate drops the Fortran order: wisetools.py:386-387);
      * fillTri / segmentTri on a region above 8192 bins (np.sum runs in buffer-sized pieces)."""
    out = {}
    rng = np.random.RandomState(4242)
    names = []
    for name, bins, order in (("p121F", [1, 2, 1], "F"), ("p12F", [1, 2], "F"), ("p1_30_1F", [1, 30, 1], "F"),
                              ("p11F", [1, 1], "F"), ("p23F", [2, 3], "F"), ("p12C", [1, 2], "C")):
        bins = np.asarray(bins, dtype=np.int64)
        data = 1.0 + 0.03 * rng.standard_normal((int(bins.sum()), 129))
        if order == "F":
            data = np.asfortranarray(data)
        with quiet(), np.errstate(all="ignore"):
            idx, dst = wt.getReference(data, list(bins), list(np.cumsum(bins)), 3, 1, 1)
        out[name + "_data"] = np.ascontiguousarray(data)
        out[name + "_fortran"] = np.bool_(order == "F")
        out[name + "_bins"] = bins
        out[name + "_idx"] = np.asarray(idx, dtype=np.int32).reshape(-1, 3)
        out[name + "_dst"] = np.asarray(dst, dtype=np.float64).reshape(-1, 3)
        names.append(name)
    out["newref_names"] = np.array(names)
    # one region of 8400 bins with a call longer than numpy's 8192-element buffer (takes minutes)
    n = 8400
    z = 0.5 * rng.standard_normal(n)
    z += 0.3                              # the whole region is one call, a few bins short of either end at most
    with np.errstate(all="ignore"):
        tri = wt.fillTri(z)
        segs = tri.segmentTri(6.0, 3)
        whole = tri.getValue(0, n - 1)
    out["long_z"] = z
    out["long_thr"] = np.float64(6.0)
    out["long_whole"] = np.float64(whole)
    out["long_seg"] = np.array([[v, x, y] for v, (x, y) in segs], dtype=np.float64).reshape(-1, 3)
    return out


# ------------------------------------------------------------- segments ----
def segment_cases(wt):
    """fillTri + TriArr.segmentTri goldens (wisetools.py:466-472, triarray.py:59-84)."""
    rng = np.random.RandomState(7)
    vecs = []
    thr = []

    def add(z, t=3.0):
        vecs.append(np.asarray(z, dtype=np.float64))
        thr.append(t)

    add([3, 0, 0, 0, 0, -3.0])                    # |min| == max -> positive wins
    add([-3, 0, 0, 0, 0, 3.0])
    add([0, 5, 0, 0, 0, 0, 5, 0.0])              # equal maxima -> lowest linear index
    add([0, 0, 0, 6, 0, 0, 0, 0, 0, 0, 0, 0.0])  # x = 3: left side not searched
    add([4, 0, 0, 0, 6, 0, 0, 0, 0, 0, 0, 0.0])  # x = 4: left side searched
    add([0, 0, 0, 0, 0, 0, 0, 6, 0, 0, 0, 4.0])  # edge 12, y = 7: right searched
    add([0, 0, 0, 0, 0, 0, 0, 0, 6, 0, 0, 4.0])  # edge 12, y = 8: right not searched
    add([1.0])
    add([5.0])
    add([-7.0, 0.5])
    add([0.0] * 9)
    add([1, 2, np.nan, 4, 5, 6, 7, 8, 9, 10.0])  # NaN window is emitted as a call
    add([0, 0, 0, 0, 0, 9, np.nan, 0, 0, 0, 0, 0, 0, 0.0])
    add([np.inf, 0, 0, 0, 0, 1, 2, 3.0])
    for n in (2, 3, 4, 5, 8, 17, 40, 130, 150):
        z = rng.standard_normal(n)
        add(z, 2.0)
        z = rng.standard_normal(n)
        lo = n // 3
        z[lo:lo + max(1, n // 4)] += 3.0
        if n > 20:
            z[-6:-2] -= 4.0
        add(z, 3.0)
    out = {"n_cases": np.int64(len(vecs)), "thresholds": np.array(thr)}
    for i, (z, t) in enumerate(zip(vecs, thr)):
        with np.errstate(all="ignore"):
            tri = wt.fillTri(z)
            segs = tri.segmentTri(t, 3)
        out["z_%d" % i] = z
        out["tri_%d" % i] = np.array(tri.data_array)
        out["seg_%d" % i] = np.array([[v, x, y] for v, (x, y) in segs], dtype=np.float64).reshape(-1, 3)
    # fillTriMin with an effect-size threshold (non-default branch, wisetools.py:479-487)
    z = rng.standard_normal(25)
    r = 1.0 + 0.05 * rng.standard_normal(25)
    z[5:12] += 3
    r[5:12] += 0.08
    with np.errstate(all="ignore"):
        tri = wt.fillTriMin(z, r, 0.05)
    out["min_z"], out["min_r"], out["min_thr"] = z, r, np.float64(0.05)
    out["min_tri"] = np.array(tri.data_array)
    # fillTriMin + segmentTri (the -mineffectsize branch), several shapes incl. NaN ratios
    cases = []
    for n, eff in ((1, 0.05), (2, 0.05), (9, 0.02), (33, 0.04), (64, 0.03), (90, 0.05), (140, 0.02)):
        z = rng.standard_normal(n)
        r = 1.0 + 0.03 * rng.standard_normal(n)
        if n > 8:
            a = n // 3
            z[a:a + n // 5] += 3.5
            r[a:a + n // 5] += 0.07
            z[-5:-1] -= 4.0
            r[-5:-1] -= 0.01      # significant z, but too small an effect: must be filtered out
        if n == 64:
            r[10] = np.nan
        cases.append((z, r, eff))
    out["mincase_n"] = np.int64(len(cases))
    for i, (z, r, eff) in enumerate(cases):
        with np.errstate(all="ignore"):
            tri = wt.fillTriMin(z, r, eff)
            segs = tri.segmentTri(3.0, 3)
        out["mincase_z_%d" % i], out["mincase_r_%d" % i], out["mincase_eff_%d" % i] = z, r, np.float64(eff)
        out["mincase_tri_%d" % i] = np.array(tri.data_array)
        out["mincase_seg_%d" % i] = np.array([[v, x, y] for v, (x, y) in segs], dtype=np.float64).reshape(-1, 3)
    return out


# ---------------------------------------------------------------- cfg1 -----
def write_sample(path, sample, binsize):
    np.savez_compressed(path, arguments={"binsize": float(binsize)}, runtime={},
                        sample=sample, quality={})


def cfg1_cases(wt, wc, n_ref=16, binsize=1000000):
    """End-to-end newref + test through the reference's own tool drivers."""
    out = {}
    profile = synth.bin_profile(binsize)
    tmp = tempfile.mkdtemp(prefix="wc_gold_")
    infiles = []
    ref_samples = []
    for i in range(n_ref):
        s = synth.make_sample(profile, seed=i)
        ref_samples.append(s)
        p = os.path.join(tmp, "ref_%02d.npz" % i)
        write_sample(p, s, binsize)
        infiles.append(p)
    keys = synth.CHROM_KEYS
    out["ref_samples"] = np.stack([np.concatenate([s[k] for k in keys]) for s in ref_samples])
    out["sample_chrom_lengths"] = np.array([len(ref_samples[0][k]) for k in keys], dtype=np.int64)
    out["binsize"] = np.float64(binsize)

    # prep -> part -> post, separately, so the prep arrays can be captured
    prep = os.path.join(tmp, "ref_prep.npz")
    with quiet(), np.errstate(all="ignore"):
        wc.toolNewrefPrep(argparse.Namespace(infiles=infiles, prepfile=prep, binsize=None))
        for part in (1, 2, 3):
            wc.toolNewrefPart(argparse.Namespace(prepfile=prep, partfile=os.path.join(tmp, "ref_part"),
                                                 part=[part, 3], refsize=100))
        refpath = os.path.join(tmp, "reference.npz")
        wc.toolNewrefPost(argparse.Namespace(prepfile=prep, partfile=os.path.join(tmp, "ref_part"),
                                             parts=3, outfile=refpath))
    pz = np.load(prep)
    for k in ("chromosomeBins", "maskedData", "mask", "maskedChromBins", "maskedChromBinSums",
              "correctedData", "pca_components", "pca_mean"):
        out["prep_" + k] = np.asarray(pz[k])
    rz = np.load(refpath)
    for k in ("indexes", "distances", "chromosome_sizes", "mask", "masked_sizes",
              "pca_components", "pca_mean"):
        out["ref_" + k] = np.asarray(rz[k])
    out["ref_binsize"] = np.float64(rz["binsize"].item())
    with np.errstate(all="ignore"):
        cutoff, _ = wt.getOptimalCutoff(rz["distances"], 3)
    out["cutoff"] = np.float64(cutoff)

    # test samples: see SURVEY.md App. A.5 for the chr5 gap cases
    events = [
        ("mild18", [("18", 20, 50, 1.05)]),
        ("gain5_gap", [("5", 40, 68, 1.5)]),     # survivors end inside the masked gap
        ("gain5_past", [("5", 40, 69, 1.5)]),
        ("gain5_after", [("5", 67, 97, 1.5)]),
        ("loss2", [("2", 100, 140, 0.5), ("11", 10, 14, 1.6)]),
        ("normal", []),
    ]
    names = []
    for j, (name, ev) in enumerate(events):
        names.append(name)
        s = synth.make_sample(profile, seed=999 + j, events=ev)
        sp = os.path.join(tmp, "test_%s.npz" % name)
        write_sample(sp, s, binsize)
        op = os.path.join(tmp, "out_%s.npz" % name)
        args = argparse.Namespace(infile=sp, outfile=op, reference=refpath, minzscore=None,
                                  chromosomes=list(range(1, 23)), mineffectsize=0, multitest=1000,
                                  minrefbins=25, repeats=5)
        with quiet(), np.errstate(all="ignore"):
            try:
                wc.toolTest(args)
            except SystemExit:
                pass
        tz = np.load(op)
        out["t_%s_sample" % name] = np.concatenate([s[k] for k in keys])
        out["t_%s_results_z" % name] = np.concatenate(list(tz["results_z"]))
        out["t_%s_results_r" % name] = np.concatenate(list(tz["results_r"]))
        out["t_%s_results_cwz" % name] = np.asarray(tz["results_cwz"], dtype=np.float64)
        out["t_%s_results_calls" % name] = np.asarray(tz["results_calls"], dtype=np.float64).reshape(-1, 5)
        for k in ("threshold_z", "asdef", "aasdef"):
            out["t_%s_%s" % (name, k)] = np.float64(tz[k])
        # function-level intermediates, from the reference's own functions
        with quiet(), np.errstate(all="ignore"):
            x = wt.toNumpyRefFormat(s, rz["chromosome_sizes"], rz["mask"])
            xp = wt.applyPCA(x, rz["pca_mean"], rz["pca_components"])
            ms = [int(v) for v in rz["masked_sizes"]]
            msum = [sum(ms[:i + 1]) for i in range(len(ms))]
            for reps in (1, 2, 5):
                z, r, n, sd = wt.repeatTest(np.copy(xp), rz["indexes"], rz["distances"], ms, msum,
                                            cutoff, float(tz["threshold_z"]), reps)
                out["t_%s_rep%d_z" % (name, reps)] = z
                out["t_%s_rep%d_n" % (name, reps)] = n
                if reps == 5:
                    out["t_%s_rep5_r" % name] = r
                    out["t_%s_rep5_sd" % name] = np.float64(sd)
        out["t_%s_x" % name] = x
        out["t_%s_xpca" % name] = xp
    out["test_names"] = np.array(names)
    # -minzscore / -chromosomes / -minrefbins / -repeats variants on one sample
    s = synth.make_sample(profile, seed=999 + 1, events=events[1][1])
    sp = os.path.join(tmp, "test_opts.npz")
    write_sample(sp, s, binsize)
    op = os.path.join(tmp, "out_opts.npz")
    args = argparse.Namespace(infile=sp, outfile=op, reference=refpath, minzscore=4.0,
                              chromosomes=[2, 5, 18], mineffectsize=0, multitest=1000,
                              minrefbins=40, repeats=2)
    with quiet(), np.errstate(all="ignore"):
        try:
            wc.toolTest(args)
        except SystemExit:
            pass
    tz = np.load(op)
    out["opts_results_z"] = np.concatenate(list(tz["results_z"]))
    out["opts_results_cwz"] = np.asarray(tz["results_cwz"], dtype=np.float64)
    out["opts_results_calls"] = np.asarray(tz["results_calls"], dtype=np.float64).reshape(-1, 5)
    out["opts_asdef"] = np.float64(tz["asdef"])
    # -mineffectsize (fillTriMin's median filter) on two samples
    for name in ("loss2", "gain5_gap"):
        sp = os.path.join(tmp, "test_%s.npz" % name)
        op = os.path.join(tmp, "outeff_%s.npz" % name)
        args = argparse.Namespace(infile=sp, outfile=op, reference=refpath, minzscore=None,
                                  chromosomes=list(range(1, 23)), mineffectsize=0.07, multitest=1000,
                                  minrefbins=25, repeats=5)
        with quiet(), np.errstate(all="ignore"):
            try:
                wc.toolTest(args)
            except SystemExit:
                pass
        tz = np.load(op)
        out["eff_%s_results_cwz" % name] = np.asarray(tz["results_cwz"], dtype=np.float64)
        out["eff_%s_results_calls" % name] = np.asarray(tz["results_calls"], dtype=np.float64).reshape(-1, 5)
    out["eff_mineffectsize"] = np.float64(0.07)
    return out



# ---------------------------------------------------------------- cfg3 -----
def cfg3_cases(wt, wc, n_ref=100, binsize=250000):
    """BASELINE configs 2 and 3 on the real reference: `newref` 100 samples x 250 kb through the
    reference's own prep / part / post drivers (about a minute), then `test` on four samples
    (about 20 s each, fillTri).  The file keeps the prep seam (correctedData, Fortran ordered like
    the reference's prep file), the reference's indexes, a SHA-256 of its distance bytes plus every
    89th distance row (the full float64 matrix would double the fixture; distances are a pure
    function of correctedData + indexes and are re-derived by the tests), and per test sample the
    function-level intermediates and the stored results."""
    import hashlib
    out = {}
    profile = synth.bin_profile(binsize)
    tmp = tempfile.mkdtemp(prefix="wc_gold3_")
    infiles = []
    for i in range(n_ref):
        p = os.path.join(tmp, "ref_%03d.npz" % i)
        write_sample(p, synth.make_sample(profile, seed=i), binsize)
        infiles.append(p)
    prep = os.path.join(tmp, "ref_prep.npz")
    refpath = os.path.join(tmp, "reference.npz")
    with quiet(), np.errstate(all="ignore"):
        wc.toolNewrefPrep(argparse.Namespace(infiles=infiles, prepfile=prep, binsize=None))
        wc.toolNewrefPart(argparse.Namespace(prepfile=prep, partfile=os.path.join(tmp, "ref_part"),
                                             part=[1, 1], refsize=100))
        wc.toolNewrefPost(argparse.Namespace(prepfile=prep, partfile=os.path.join(tmp, "ref_part"),
                                             parts=1, outfile=refpath))
    pz = np.load(prep)
    corrected = pz["correctedData"]
    assert corrected.flags["F_CONTIGUOUS"] and not corrected.flags["C_CONTIGUOUS"]
    out["prep_correctedData"] = np.ascontiguousarray(corrected)     # values; the layout flag is below
    out["prep_fortran"] = np.bool_(True)
    out["prep_maskedChromBins"] = np.asarray(pz["maskedChromBins"], dtype=np.int64)
    rz = np.load(refpath)
    dist = np.ascontiguousarray(rz["distances"], dtype=np.float64)
    out["ref_indexes"] = np.asarray(rz["indexes"], dtype=np.int32)
    out["ref_distances_sha256"] = np.array(hashlib.sha256(dist.tobytes()).hexdigest())
    out["ref_distance_rows"] = np.arange(0, dist.shape[0], 89, dtype=np.int64)
    out["ref_distances_sampled"] = dist[::89].copy()
    for k in ("chromosome_sizes", "mask", "masked_sizes", "pca_components", "pca_mean"):
        out["ref_" + k] = np.asarray(rz[k])
    out["ref_binsize"] = np.float64(rz["binsize"].item())
    with np.errstate(all="ignore"):
        cutoff, _ = wt.getOptimalCutoff(rz["distances"], 3)
    out["cutoff"] = np.float64(cutoff)

    keys = synth.CHROM_KEYS
    out["sample_chrom_lengths"] = np.array([len(p) for p in profile], dtype=np.int64)
    events = [
        ("mild18", [("18", 100, 220, 1.05)]),                       # oracle calls [18, 98, 217]
        ("strong5", [("5", 200, 262, 1.5), ("13", 150, 190, 0.5)]),  # flags change reference sets across repeats
        ("loss2", [("2", 300, 420, 0.93), ("11", 40, 44, 1.8)]),
        ("normal", []),
    ]
    names = []
    ms = [int(v) for v in rz["masked_sizes"]]
    msum = [sum(ms[:i + 1]) for i in range(len(ms))]
    for j, (name, ev) in enumerate(events):
        names.append(name)
        s = synth.make_sample(profile, seed=999 + j, events=ev)
        sp = os.path.join(tmp, "test_%s.npz" % name)
        write_sample(sp, s, binsize)
        op = os.path.join(tmp, "out_%s.npz" % name)
        args = argparse.Namespace(infile=sp, outfile=op, reference=refpath, minzscore=None,
                                  chromosomes=list(range(1, 23)), mineffectsize=0, multitest=1000,
                                  minrefbins=25, repeats=5)
        with quiet(), np.errstate(all="ignore"):
            try:
                wc.toolTest(args)
            except SystemExit:
                pass
        tz = np.load(op)
        out["t_%s_sample" % name] = np.concatenate([s[k] for k in keys])
        out["t_%s_results_z" % name] = np.concatenate(list(tz["results_z"]))
        out["t_%s_results_r" % name] = np.concatenate(list(tz["results_r"]))
        out["t_%s_results_cwz" % name] = np.asarray(tz["results_cwz"], dtype=np.float64)
        out["t_%s_results_calls" % name] = np.asarray(tz["results_calls"], dtype=np.float64).reshape(-1, 5)
        for k in ("threshold_z", "asdef", "aasdef"):
            out["t_%s_%s" % (name, k)] = np.float64(tz[k])
        with quiet(), np.errstate(all="ignore"):
            x = wt.toNumpyRefFormat(s, rz["chromosome_sizes"], rz["mask"])
            xp = wt.applyPCA(x, rz["pca_mean"], rz["pca_components"])
            for reps in (1, 5):
                z, r, n, sd = wt.repeatTest(np.copy(xp), rz["indexes"], rz["distances"], ms, msum,
                                            cutoff, float(tz["threshold_z"]), reps)
                out["t_%s_rep%d_z" % (name, reps)] = z
                out["t_%s_rep%d_n" % (name, reps)] = np.asarray(n, dtype=np.int16)
                if reps == 5:
                    out["t_%s_rep5_r" % name] = r
                    out["t_%s_rep5_sd" % name] = np.float64(sd)
        out["t_%s_xpca" % name] = xp
        print(name, "calls:", out["t_%s_results_calls" % name][:, :3].tolist(), flush=True)
    out["test_names"] = np.array(names)
    return out


# ---------------------------------------------------------------- cfg5 -----
def cfg5_cases(wt):
    """BASELINE config 5 (50 kb): the real reference's fillTri + segmentTri on the three longest
    chromosomes of one sample.  The z vectors come from this repo's GPU
    path on the deterministic case of tools/cfg5_case.py (gpurun_out/cfg5_z.npz, written on the
    GPU box by tools/gpu_cfg5_z.py); ~11 M windows per chromosome, a few minutes of np.sum."""
    src = np.load(os.path.join(ROOT, "gpurun_out", "cfg5_z.npz"))
    # only INPUTS of the reference run come from the GPU box (the cleaned z / ratio vectors); what the HIP
    # `test` path called there (calls_sample0, n_calls_all in gpurun_out/cfg5_z.npz) is not copied: whole
    # samples are pinned by tests/golden/cfg5_whole.npz (tools/make_cfg5_whole.py) instead
    out = {"threshold": np.float64(src["threshold"]), "masked_bins": src["masked_bins"]}
    thr = float(src["threshold"])
    for c in (1, 2, 3):
        z = np.asarray(src["z_chr%d" % c], dtype=np.float64)
        with np.errstate(all="ignore"):
            tri = wt.fillTri(z)
            segs = tri.segmentTri(thr, 3)
            whole = tri.getValue(0, len(z) - 1)
        out["z_chr%d" % c] = z
        out["r_chr%d" % c] = np.asarray(src["r_chr%d" % c], dtype=np.float64)
        out["whole_chr%d" % c] = np.float64(whole)
        out["seg_chr%d" % c] = np.array([[v, x, y] for v, (x, y) in segs], dtype=np.float64).reshape(-1, 3)
        # spot values of the triangle itself (the packed array is 11 M doubles: not stored)
        rng = np.random.RandomState(c)
        xs = rng.randint(0, len(z), size=400)
        ys = np.array([rng.randint(x, len(z)) for x in xs])
        out["tri_x_chr%d" % c], out["tri_y_chr%d" % c] = xs.astype(np.int32), ys.astype(np.int32)
        out["tri_v_chr%d" % c] = np.array([tri.getValue(int(x), int(y)) for x, y in zip(xs, ys)])
        print("chr%d" % c, len(z), "segments", out["seg_chr%d" % c][:, 1:].tolist(),
              "gpu said", src["gpu_seg_chr%d" % c][:, 1:].tolist(), flush=True)
    return out


# ---------------------------------------------------------- refsize 300 -----
def refsize300_cases(wt, wc):
    """`-refsize 300` (beyond one numpy pairwise block in `test`, beyond the candidate lists' design
    size in `newref`): the reference's newrefpart / newrefpost / test on the 1 Mb prep seam of
    cfg1_pipeline.npz."""
    import hashlib
    g = np.load(os.path.join(GOLD, "cfg1_pipeline.npz"))
    tmp = tempfile.mkdtemp(prefix="wc_gold300_")
    prep = os.path.join(tmp, "ref_prep.npz")
    corrected = np.asfortranarray(g["prep_correctedData"])
    ms = [int(v) for v in g["prep_maskedChromBins"]]
    np.savez_compressed(prep, binsize=np.float64(g["binsize"]), chromosomeBins=g["prep_chromosomeBins"],
                        maskedData=g["prep_maskedData"], mask=g["prep_mask"], maskedChromBins=ms,
                        maskedChromBinSums=[sum(ms[:i + 1]) for i in range(len(ms))], correctedData=corrected,
                        pca_components=g["prep_pca_components"], pca_mean=g["prep_pca_mean"], arguments={}, runtime={})
    assert np.load(prep)["correctedData"].flags["F_CONTIGUOUS"]
    refpath = os.path.join(tmp, "reference.npz")
    with quiet(), np.errstate(all="ignore"):
        wc.toolNewrefPart(argparse.Namespace(prepfile=prep, partfile=os.path.join(tmp, "ref_part"), part=[1, 1], refsize=300))
        wc.toolNewrefPost(argparse.Namespace(prepfile=prep, partfile=os.path.join(tmp, "ref_part"), parts=1, outfile=refpath))
    rz = np.load(refpath)
    dist = np.ascontiguousarray(rz["distances"], dtype=np.float64)
    out = {"k": np.int64(300), "ref_indexes": np.asarray(rz["indexes"], dtype=np.int32),
           "ref_distances_sha256": np.array(hashlib.sha256(dist.tobytes()).hexdigest()),
           "ref_distance_rows": np.arange(0, dist.shape[0], 97, dtype=np.int64), "ref_distances_sampled": dist[::97].copy()}
    with np.errstate(all="ignore"):
        cutoff, _ = wt.getOptimalCutoff(rz["distances"], 3)
    out["cutoff"] = np.float64(cutoff)
    keys = synth.CHROM_KEYS
    lengths = g["sample_chrom_lengths"]
    offs = np.concatenate([[0], np.cumsum(lengths)])
    msum = [sum(ms[:i + 1]) for i in range(len(ms))]
    for name in ("gain5_gap", "loss2"):
        flat = g["t_%s_sample" % name]
        s = {k: np.asarray(flat[offs[i]:offs[i + 1]], dtype=np.int32) for i, k in enumerate(keys)}
        sp = os.path.join(tmp, "test_%s.npz" % name)
        write_sample(sp, s, float(g["binsize"]))
        op = os.path.join(tmp, "out_%s.npz" % name)
        args = argparse.Namespace(infile=sp, outfile=op, reference=refpath, minzscore=None,
                                  chromosomes=list(range(1, 23)), mineffectsize=0, multitest=1000,
                                  minrefbins=25, repeats=5)
        with quiet(), np.errstate(all="ignore"):
            try:
                wc.toolTest(args)
            except SystemExit:
                pass
        tz = np.load(op)
        out["t_%s_results_z" % name] = np.concatenate(list(tz["results_z"]))
        out["t_%s_results_cwz" % name] = np.asarray(tz["results_cwz"], dtype=np.float64)
        out["t_%s_results_calls" % name] = np.asarray(tz["results_calls"], dtype=np.float64).reshape(-1, 5)
        out["t_%s_asdef" % name] = np.float64(tz["asdef"])
        with quiet(), np.errstate(all="ignore"):
            z, r, n, sd = wt.repeatTest(np.copy(g["t_%s_xpca" % name]), rz["indexes"], rz["distances"], ms, msum,
                                        cutoff, float(tz["threshold_z"]), 5)
        out["t_%s_rep5_z" % name], out["t_%s_rep5_r" % name] = z, r
        out["t_%s_rep5_n" % name], out["t_%s_rep5_sd" % name] = np.asarray(n, dtype=np.int16), np.float64(sd)
        print(name, "max refs", int(np.max(n)), "calls", out["t_%s_results_calls" % name][:, :3].tolist(), flush=True)
    return out

# ------------------------------------------------------------- binsize -----
def scale_cases(wt):
    """scaleSample + a 2-sample newrefprep at a merged bin size (wisetools.py:220-264)."""
    rng = np.random.RandomState(3)
    sample = {k: rng.poisson(30, size=n).astype(np.int32)
              for k, n in zip(synth.CHROM_KEYS, rng.randint(7, 30, size=24))}
    with quiet():
        scaled = wt.scaleSample(sample, 50000., 250000)
    out = {"lengths": np.array([len(sample[k]) for k in synth.CHROM_KEYS]),
           "sample": np.concatenate([sample[k] for k in synth.CHROM_KEYS]),
           "scaled_lengths": np.array([len(scaled[k]) for k in synth.CHROM_KEYS]),
           "scaled": np.concatenate([scaled[k] for k in synth.CHROM_KEYS])}
    return out


# ------------------------------------------------- cfg4 slice (600 x 50 kb) --
def cfg4slice_rows(bins):
    """64 target rows of the 600 x 50 kb matrix: first / last bin of several chromosomes, bins inside
    chr1, chr21 and chr22, and a seeded handful anywhere."""
    bins = np.asarray(bins, dtype=np.int64)
    ends = np.cumsum(bins)
    starts = ends - bins
    rows = []
    for c in (0, 1, 6, 11, 17, 20, 21):
        rows += [int(starts[c]), int(ends[c] - 1)]
    rows += [int(starts[0] + v) for v in (1, 2, 777, 2492, 4000)]
    rows += [int(starts[20] + v) for v in (1, 100, 481, 900)]
    rows += [int(starts[21] + v) for v in (1, 313, 1000)]
    rng = np.random.RandomState(4)
    while len(rows) < 64:
        r = int(rng.randint(0, int(ends[-1])))
        if r not in rows:
            rows.append(r)
    return np.array(sorted(rows), dtype=np.int64)


def cfg4slice_cases(wt):
    """BASELINE config 4 (600 samples x 50 kb, 57 633 bins): the REAL reference's getReference
    (wisetools.py:364-398, getRefForBins :298-325) for 64 target rows against ALL candidates, on the
    kernel-level matrix in C order (numpy sums each row pairwise) and in Fortran order (the prep file's
    layout: sample by sample).  `part = row + 1 of B parts` makes getPart return exactly [row, row + 1)
    (B / float(B) is 1.0).  Only the rows, indexes and distances are stored; the matrix is
    synth.corrected_matrix(50000, 600, seed=0)."""
    data, bins, sums = synth.corrected_matrix(50000, 600, seed=0)
    B = int(sums[-1])
    rows = cfg4slice_rows(bins)
    out = {"rows": rows, "shape": np.array(data.shape, dtype=np.int64), "k": np.int64(100),
           "data_probe": data[rows[:4], :3].copy()}
    for tag, lay in (("c", np.ascontiguousarray(data)), ("f", np.asfortranarray(data))):
        idx = np.empty((len(rows), 100), dtype=np.int32)
        dst = np.empty((len(rows), 100), dtype=np.float64)
        for n, row in enumerate(rows):
            with quiet(), np.errstate(all="ignore"):
                i_, d_ = wt.getReference(lay, list(bins), list(sums), 100, int(row) + 1, B)
            i_ = np.asarray(i_); d_ = np.asarray(d_)
            assert i_.shape == (1, 100), i_.shape
            idx[n], dst[n] = i_[0], d_[0]
        out["idx_" + tag], out["dst_" + tag] = idx, dst
        del lay
    return out


def cutoff_mask_cases(wt):
    """getOptimalCutoff's BOTH return values (wisetools.py:328-336) from the real reference, on the
    cfg1 reference's distances (read from the committed cfg1 fixture) and on a tiny hand-made array."""
    g = np.load(os.path.join(GOLD, "cfg1_pipeline.npz"), allow_pickle=False)
    d = np.asarray(g["ref_distances"], dtype=np.float64)
    small = np.sort(np.random.RandomState(8).gamma(3.0, 0.1, size=(40, 9)), axis=1)
    small[::7, -2:] = 1e10
    out = {"small": small}
    for name, arr in (("cfg1", d), ("small", small)):
        for repeats in (0, 1, 2, 3):
            with quiet(), np.errstate(all="ignore"):
                cut, mask = wt.getOptimalCutoff(arr, repeats)
            out["%s_cutoff_%d" % (name, repeats)] = np.float64(cut)
            out["%s_mask_%d" % (name, repeats)] = np.packbits(np.asarray(mask).astype(bool).ravel())
            out["%s_maskdtype_%d" % (name, repeats)] = np.array(str(np.asarray(mask).dtype))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None,
                    help="regenerate one file only: layout | cfg3 | cfg5 | refsize300 | cfg4slice | cutoffmask")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    wt, wc, _tri = ref_loader.load(full_svd=True)
    save = ref_loader.np.savez_compressed
    if args.only in (None, "layout"):
        save(os.path.join(GOLD, "layout_cases.npz"), **layout_cases(wt))
    if args.only in (None, "cfg3"):
        save(os.path.join(GOLD, "cfg3_250kb.npz"), **cfg3_cases(wt, wc))
    if args.only == "refsize300":
        save(os.path.join(GOLD, "refsize300.npz"), **refsize300_cases(wt, wc))
    if args.only == "cutoffmask":
        save(os.path.join(GOLD, "cutoff_mask.npz"), **cutoff_mask_cases(wt))
    if args.only == "cfg4slice":
        save(os.path.join(GOLD, "cfg4slice.npz"), **cfg4slice_cases(wt))
    if args.only == "cfg5":
        save(os.path.join(GOLD, "cfg5_50kb.npz"), **cfg5_cases(wt))
    if args.only is not None:
        return
    save(os.path.join(GOLD, "newref_kernel.npz"), **newref_cases(wt))
    save(os.path.join(GOLD, "segments.npz"), **segment_cases(wt))
    save(os.path.join(GOLD, "scale.npz"), **scale_cases(wt))
    save(os.path.join(GOLD, "cfg1_pipeline.npz"), **cfg1_cases(wt, wc))
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
