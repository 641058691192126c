"""The deterministic 50 kb case behind tests/golden/cfg5_50kb.npz (BASELINE config 5's per-GPU
share: 125 samples x 50 kb bins), shared by the script that produced the golden's inputs on the
GPU box (tools/gpu_cfg5_z.py) and by tests/test_cfg5_gpu.py.

A reference cannot be built at 50 kb with the real reference in reasonable time (about 3.5 core
hours), so the reference set is built by this repo on the GPU (40 synthetic samples,
prep + newref -- both verified against the real reference at 1 Mb and 250 kb); what the REAL
reference contributes at this size is the part that dominates `test`: fillTri + segmentTri on
sample 0's three longest chromosomes (tools/make_goldens.py --only cfg5).
"""
import numpy as np

BINSIZE = 50000
N_REF = 40
N_TEST = 125
GOLDEN_CHROMS = (0, 1, 2)        # chromosomes 1-3: 4986, 4864, 3961 bins before masking


def test_samples(synth, profile, n_test=N_TEST):
    """The test cohort (no GPU needed): sample 0 carries a gain on chromosome 1 and a loss on
    chromosome 2 (chromosome 3 stays clean); every fifth other sample a mild event."""
    rng = np.random.RandomState(77)
    tests = []
    for i in range(n_test):
        events = []
        if i == 0:
            events = [("1", 1500, 2100, 1.04), ("2", 3000, 3400, 0.95)]
        elif i % 5 == 0:
            c = int(rng.randint(1, 23))
            n = len(profile[c - 1])
            a = int(rng.randint(0, n - n // 5))
            events = [(str(c), a, a + n // 5, 1.0 + rng.choice([-1, 1]) * rng.uniform(0.02, 0.05))]
        tests.append(synth.make_sample(profile, seed=5000 + i, events=events))
    return tests


def build(wt, synth, n_test=N_TEST):
    """Reference (device resident) + the test cohort.  Sample 0 carries a gain on chromosome 1
    and a loss on chromosome 2 (chromosome 3 stays clean); every fifth other sample a mild event."""
    from wisecondor_amd.wisecondor import zThreshold
    profile = synth.bin_profile(BINSIZE)
    refs = [synth.make_sample(profile, seed=i) for i in range(N_REF)]
    _, chrom_bins, mask, corrected, comps, mean, masked_bins = wt.prepReference(refs)
    masked_bins = np.asarray(masked_bins, dtype=np.int64)
    idx, dst = wt.getReference(corrected, masked_bins, np.cumsum(masked_bins), 100, 1, 1)
    reference = wt.Reference(idx, dst, np.asarray(chrom_bins, dtype=np.int64), masked_bins, mask, mean, comps,
                             binsize=BINSIZE, device=0)
    thr = float(zThreshold([int(v) for v in masked_bins], 1000, None))
    tests = test_samples(synth, profile, n_test)
    return dict(reference=reference, threshold=thr, tests=tests, masked_bins=masked_bins,
                chrom_bins=np.asarray(chrom_bins, dtype=np.int64), indexes=idx, distances=dst,
                corrected=corrected)


def cleaned_regions(wt, case, sample_index=0, minrefbins=25):
    """(z, r) per chromosome of one sample after the minrefbins cleaning (wisecondor.py:215-222):
    prepare -> 5 z-score repeats -> keep bins with refSizes >= minrefbins."""
    from wisecondor_amd import _lib
    ref = case["reference"]
    counts = wt.samples_to_counts([case["tests"][sample_index]], case["chrom_bins"])
    data = np.empty((1, ref.n_bins))
    _lib.check(_lib.load().wc_prepare_samples(ref.ctx, ref.handle, _lib.ptr(counts), 1, _lib.ptr(data), None))
    z, r, n, sd = wt.repeatTest(data[0], None, None, None, None, None, case["threshold"], 5, reference=ref)
    keep = n >= minrefbins
    offs = np.concatenate([[0], np.cumsum(case["masked_bins"])])
    zs = [z[offs[c]:offs[c + 1]][keep[offs[c]:offs[c + 1]]] for c in range(22)]
    rs = [r[offs[c]:offs[c + 1]][keep[offs[c]:offs[c + 1]]] for c in range(22)]
    return zs, rs, data[0], (z, r, n, sd)
