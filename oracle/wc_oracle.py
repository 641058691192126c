"""CPU oracle for the WISECONDOR `newref` / `test` hot path.

TEST INFRASTRUCTURE ONLY.  This module is a numpy float64 restatement of the
reference algorithm, written from the behavioural spec in SURVEY.md App. A and
checked against golden vectors that were produced by running the real
reference (tools/make_goldens.py -> tests/golden/*.npz).  It may be imported
only by tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline`
leg -- never by the product package `wisecondor_amd`, which must fail loudly
when its HIP library is missing.

Parity status: the reference ships no tests or golden files of its own
(SURVEY.md section 4), so this oracle is pinned against outputs of the
reference itself, run in the development container under numpy 2.2.6 /
scikit-learn 1.7.2 / scipy 1.15.3 (tests/test_oracle_vs_golden.py).

Every function cites the reference lines (file:line under /root/reference) it
restates.  The structure deliberately keeps the reference's cost model (one
numpy temporary per target bin, one np.sum per Stouffer window) so that it can
double as the "port" CPU baseline.
"""
import bisect

import numpy as np
from scipy.stats import norm

_ERR = dict(divide="ignore", invalid="ignore", over="ignore", under="ignore")

SENTINEL_INDEX = -1      # wisetools.py:305
SENTINEL_DISTANCE = 1e10  # wisetools.py:306


# --------------------------------------------------------------------------
# numpy's float64 pairwise summation, spelled out.  The HIP kernels reproduce
# this order so that re-scored distances, means and standard deviations carry
# numpy's bits.  (numpy/core/src/umath/loops_utils.h.src, pairwise_sum; call
# sites in the reference: wisetools.py:302, 426-427, 471.)
# --------------------------------------------------------------------------
NPY_BUFSIZE = 8192      # np.getbufsize(): add.reduce hands the inner loop at most this many elements


def pairwise_sum(a):
    """Sum a 1-D float64 sequence in numpy's add.reduce order: pairwise within pieces of
    NPY_BUFSIZE elements, the piece sums accumulated left to right."""
    a = np.asarray(a, dtype=np.float64)
    if a.shape[0] > NPY_BUFSIZE:
        res = 0.0
        for off in range(0, a.shape[0], NPY_BUFSIZE):
            res = res + _pairwise_piece(a[off:off + NPY_BUFSIZE])
        return res
    return _pairwise_piece(a)


def _pairwise_piece(a):
    """numpy's pairwise_sum (loops_utils.h.src) of one piece."""
    n = a.shape[0]
    if n < 8:
        res = 0.0
        for v in a:
            res = res + float(v)
        return res
    if n <= 128:
        r = [float(a[j]) for j in range(8)]
        body = n - (n % 8)
        for i in range(8, body, 8):
            for j in range(8):
                r[j] = r[j] + float(a[i + j])
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for i in range(body, n):
            res = res + float(a[i])
        return res
    half = n // 2
    half -= half % 8
    return _pairwise_piece(a[:half]) + _pairwise_piece(a[half:])


# --------------------------------------------------------------------------
# newref: reference-bin selection
# --------------------------------------------------------------------------
def get_part(partnum, outof, bincount):
    """Rows [start, end) of zero-based part `partnum` of `outof` (wisetools.py:358-361)."""
    start = int(bincount / float(outof) * partnum)
    end = int(bincount / float(outof) * (partnum + 1))
    return start, end


def split_by_chrom(start, end, chrom_bin_sums):
    """Cut [start, end) at chromosome ends (wisetools.py:340-354).

    Returns [chrom, region_start, region_end] triples.  As in the reference the
    first triple's start can lie before `start`; get_reference clamps it.
    """
    regions = []
    cur_chrom, cur_start = 0, start
    for chrom, cum in enumerate(chrom_bin_sums):
        cur_chrom = chrom
        if cum >= end:
            break
        if start < cum < end:
            regions.append([cur_chrom, cur_start, int(cum)])
            cur_chrom, cur_start = chrom, int(cum)
        cur_start = int(cum)
    regions.append([cur_chrom, cur_start, end])
    return regions


def get_ref_for_bins(amount, start, end, sample_data, other_data):
    """Top-`amount` nearest rows of `other_data` for target rows [start, end).

    wisetools.py:298-325.  Distance is sum over samples of squared differences
    (:302); selection is an ascending insertion with bisect_right guarded by a
    strict `<` against the current k-th value (:313-321), i.e. a stable
    (distance, position) order with -1 / 1e10 padding.
    """
    ref_indexes = np.zeros((end - start, amount), dtype=np.int32)
    ref_distances = np.ones((end - start, amount))
    for this_bin in range(start, end):
        dist = np.sum(np.power(other_data - sample_data[this_bin, :], 2), 1)
        best_idx = [SENTINEL_INDEX] * amount
        best_dst = [SENTINEL_DISTANCE] * amount
        cur_max = SENTINEL_DISTANCE
        for pos, val in enumerate(dist):
            if val < cur_max:
                at = bisect.bisect(best_dst, val)
                best_idx.pop()
                best_dst.pop()
                best_idx.insert(at, pos)
                best_dst.insert(at, val)
                cur_max = best_dst[-1]
        ref_indexes[this_bin - start, :] = best_idx
        ref_distances[this_bin - start, :] = best_dst
    return ref_indexes, ref_distances


def get_ref_for_bins_fast(amount, start, end, sample_data, other_data):
    """Same result as get_ref_for_bins via a stable argsort.

    Used where the pure-Python insertion scan would make a test too slow; the
    equivalence is itself asserted in tests/test_oracle_vs_golden.py.
    """
    rows = end - start
    ref_indexes = np.full((rows, amount), SENTINEL_INDEX, dtype=np.int32)
    ref_distances = np.full((rows, amount), SENTINEL_DISTANCE)
    for this_bin in range(start, end):
        dist = np.sum(np.power(other_data - sample_data[this_bin, :], 2), 1)
        order = np.argsort(dist, kind="stable")
        order = order[dist[order] < SENTINEL_DISTANCE][:amount]  # NaN / >=1e10 never admitted
        ref_indexes[this_bin - start, :order.shape[0]] = order
        ref_distances[this_bin - start, :order.shape[0]] = dist[order]
    return ref_indexes, ref_distances


def get_reference(corrected, chrom_bins, chrom_bin_sums, select_ref_amount=100,
                  part=1, split_parts=1, fast=False):
    """Reference bins for the rows of part `part` (1-based) of `split_parts`.

    wisetools.py:364-398.  Candidates of a target on chromosome c are all rows
    not on c (:386-387); stored indexes are positions in that concatenation.
    """
    kernel = get_ref_for_bins_fast if fast else get_ref_for_bins
    bincount = int(chrom_bin_sums[-1])
    start_num, end_num = get_part(part - 1, split_parts, bincount)
    big_idx, big_dst = [], []
    for chrom, start, end in split_by_chrom(start_num, end_num, chrom_bin_sums):
        start = max(start, start_num)
        end = min(end, end_num)
        lo = int(chrom_bin_sums[chrom] - chrom_bins[chrom])
        hi = int(chrom_bin_sums[chrom])
        chrom_data = np.concatenate((corrected[:lo, :], corrected[hi:, :]))
        idx, dst = kernel(select_ref_amount, start, end, corrected, chrom_data)
        big_idx.extend(idx)
        big_dst.extend(dst)
    return np.array(big_idx), np.array(big_dst)


# --------------------------------------------------------------------------
# newref prep (upstream of the hot path; needed to drive the CLI end to end)
# --------------------------------------------------------------------------
def scale_sample(sample, from_size, to_size):
    """Merge bins to a coarser size (wisetools.py:220-237)."""
    if to_size is None or from_size == to_size:
        return sample
    if to_size == 0 or from_size == 0 or to_size < from_size or to_size % from_size > 0:
        raise ValueError("Impossible binsize scaling requested: %s to %s" % (from_size, to_size))
    scale = int(to_size / from_size)
    out = {}
    for chrom, data in sample.items():
        new_len = int(np.ceil(len(data) / float(scale)))
        scaled = np.zeros(new_len, dtype=np.int32)
        for i in range(new_len):
            scaled[i] = np.sum(data[i * scale:i * scale + scale])
        out[chrom] = scaled
    return out


def to_numpy_array(samples):
    """Samples -> masked, unit-sum [bins, samples] matrix (wisetools.py:240-264)."""
    by_chrom, chrom_bins = [], []
    for chrom in range(1, 23):
        max_len = max(s[str(chrom)].shape[0] for s in samples)
        block = np.zeros((max_len, len(samples)), dtype=float)
        chrom_bins.append(max_len)
        for col, s in enumerate(samples):
            block[:, col] = s[str(chrom)]
        by_chrom.append(block)
    all_data = np.concatenate(by_chrom, axis=0)
    with np.errstate(**_ERR):
        all_data = all_data / np.sum(all_data, 0)
    mask = np.sum(all_data, 1) > 0
    return all_data[mask, :], chrom_bins, mask


def train_pca(ref_data, pcacomp=3):
    """Exact rank-`pcacomp` PCA correction (wisetools.py:89-101).

    Restated as a deterministic full SVD of the centred [samples, bins]
    matrix with the sign convention of scikit-learn 1.7 (svd_flip on the rows
    of Vt), so that it matches svd_solver='full' (SURVEY.md section 8f rank 1).
    """
    t = ref_data.T
    mean = np.mean(t, axis=0)
    centred = t - mean
    u, s, vt = np.linalg.svd(centred, full_matrices=False)
    max_abs = np.argmax(np.abs(vt), axis=1)
    signs = np.sign(vt[range(vt.shape[0]), max_abs])
    vt = vt * signs[:, np.newaxis]
    comps = vt[:pcacomp]
    transformed = np.dot(centred, comps.T)
    inversed = np.dot(transformed, comps) + mean
    with np.errstate(**_ERR):
        corrected = t / inversed
    return corrected.T, comps, mean


# --------------------------------------------------------------------------
# test: sample preparation
# --------------------------------------------------------------------------
def to_numpy_ref_format(sample, chrom_bins, mask):
    """Pad/truncate to the reference layout, normalise, mask (wisetools.py:267-278)."""
    by_chrom = []
    for chrom in range(1, 23):
        want = int(chrom_bins[chrom - 1])
        block = np.zeros(want, dtype=float)
        have = min(want, len(sample[str(chrom)]))
        block[:have] = sample[str(chrom)][:have]
        by_chrom.append(block)
    all_data = np.concatenate(by_chrom, axis=0)
    with np.errstate(**_ERR):
        all_data = all_data / np.sum(all_data)
    return all_data[mask]


def apply_pca(sample_data, mean, components):
    """x / reconstruction from the stored components (wisetools.py:104-113)."""
    transform = np.dot(np.array([sample_data]) - mean, components.T)
    reconstructed = (np.dot(transform, components) + mean)[0]
    with np.errstate(**_ERR):
        return sample_data / reconstructed


def get_optimal_cutoff(distances, repeats):
    """Iterated mean + 3 sd clip of the reference distances (wisetools.py:328-336)."""
    cutoff = float("inf")
    mask = np.zeros(distances.shape)
    for _ in range(repeats):
        mask = distances < cutoff
        average = np.average(distances[mask])
        stddev = np.std(distances[mask])
        cutoff = average + 3 * stddev
    return cutoff, mask


def z_threshold(masked_sizes, multitest=1000, minzscore=None):
    """Per-bin threshold (wisecondor.py:203-207)."""
    if minzscore is not None:
        return minzscore
    num_tests = sum(masked_sizes)
    return norm.ppf(1 - 1. / (num_tests * 0.5 * multitest))


# --------------------------------------------------------------------------
# test: per-bin z-scores with iterative masking
# --------------------------------------------------------------------------
def try_sample(test_data, test_copy, indexes, distances, chrom_bins, chrom_bin_sums, cutoff):
    """One z-score pass (wisetools.py:407-435)."""
    bincount = int(chrom_bin_sums[-1])
    z = np.zeros(bincount)
    r = np.zeros(bincount)
    ref_sizes = np.zeros(bincount)
    sd_sum, sd_num = 0., 0
    i = 0
    with np.errstate(**_ERR):
        for chrom in range(len(chrom_bins)):
            lo = int(chrom_bin_sums[chrom] - chrom_bins[chrom])
            hi = int(chrom_bin_sums[chrom])
            others = np.concatenate((test_copy[:lo], test_copy[hi:]))
            for index in indexes[lo:hi]:
                ref = others[index[distances[i] < cutoff]]
                ref = ref[ref >= 0]
                if ref.shape[0]:
                    mean, sd = np.mean(ref), np.std(ref)
                else:
                    mean, sd = np.nan, np.nan
                if not np.isnan(sd):
                    sd_sum += sd
                    sd_num += 1
                z[i] = (test_data[i] - mean) / sd
                r[i] = test_data[i] / mean
                ref_sizes[i] = ref.shape[0]
                i += 1
        sd_avg = sd_sum / sd_num if sd_num else np.nan
    return z, r, ref_sizes, sd_avg


def repeat_test(test_data, indexes, distances, chrom_bins, chrom_bin_sums, cutoff, threshold, repeats):
    """`repeats` passes, flagging |z| >= threshold bins as -1 (wisetools.py:438-448)."""
    copy = np.copy(test_data)
    out = None
    for _ in range(repeats):
        out = try_sample(test_data, copy, indexes, distances, chrom_bins, chrom_bin_sums, cutoff)
        with np.errstate(**_ERR):
            copy[np.abs(out[0]) >= threshold] = -1
    return out


# --------------------------------------------------------------------------
# test: Stouffer window triangle and segmentation
# --------------------------------------------------------------------------
def tri_size(edge):
    return int((edge * edge) / 2. + edge / 2.)  # triarray.py:17


def tri_offset(edge, x, y):
    """Linear position of window (x, y) in the packed triangle (triarray.py:28-29)."""
    return x * edge - (x * (x - 1)) // 2 + y - x


def lin_to_2d(edge, pos):
    """Inverse of tri_offset (triarray.py:46-51)."""
    cur = edge
    while pos >= cur:
        pos -= cur
        cur -= 1
    return edge - cur, pos + edge - cur


def fill_tri(region):
    """Packed triangle of sum(z[x..y]) / sqrt(y-x+1) (wisetools.py:466-472)."""
    n = region.shape[0]
    tri = np.zeros(tri_size(n))
    at = 0
    with np.errstate(**_ERR):
        for x in range(n):
            for y in range(x, n):
                tri[at] = np.sum(region[x:y + 1]) / np.sqrt(y - x + 1)
                at += 1
    return tri


def fill_tri_min(region_z, region_r, threshold):
    """fill_tri with windows of small median effect zeroed (wisetools.py:475-487)."""
    if threshold == 0:
        return fill_tri(region_z)
    n = region_z.shape[0]
    tri = np.zeros(tri_size(n))
    at = 0
    with np.errstate(**_ERR):
        for x in range(n):
            for y in range(x, n):
                if abs(np.median(region_r[x:y + 1]) - 1) >= threshold:
                    tri[at] = np.sum(region_z[x:y + 1]) / np.sqrt(y - x + 1)
                at += 1
    return tri


def _sub_triangle(tri, edge, start, end):
    """Copy of the windows inside [start, end) (triarray.py:31-38)."""
    sub_edge = end - start
    sub = np.zeros(tri_size(sub_edge))
    at = 0
    for x in range(start, end):
        base = tri_offset(edge, x, x)
        cnt = end - x
        sub[at:at + cnt] = tri[base:base + cnt]
        at += cnt
    return sub, sub_edge


def segment_tri(tri, edge, threshold, min_search=3):
    """Recursive most-significant-segment calling (triarray.py:59-84).

    Returns [(value, (x, y))] with inclusive window bounds, ascending.
    """
    out = []
    if tri.shape[0] == 0:
        return out
    champ_pos = int(tri.argmax())
    champ_val = tri[champ_pos]
    bot_pos = int(tri.argmin())
    bot_val = tri[bot_pos]
    if abs(bot_val) > champ_val:
        champ_val, champ_pos = bot_val, bot_pos
    if abs(champ_val) < threshold:
        return out
    x, y = lin_to_2d(edge, champ_pos)
    if x > min_search:
        sub, sub_edge = _sub_triangle(tri, edge, 0, x)
        out.extend(segment_tri(sub, sub_edge, threshold, min_search))
    out.append((champ_val, (x, y)))
    if y + 1 < edge - min_search:
        sub, sub_edge = _sub_triangle(tri, edge, y + 1, edge)
        right = segment_tri(sub, sub_edge, threshold, min_search)
        out.extend((v, (a + y + 1, b + y + 1)) for v, (a, b) in right)
    return out


def inflate_array(array, mask):
    """Scatter `array` into the True positions of `mask` (wisetools.py:281-288)."""
    out = np.zeros(mask.shape[0])
    out[np.flatnonzero(mask)] = array
    return out


def inflate_array_multi(array, mask_list):
    """Undo nested maskings, innermost last (wisetools.py:291-295)."""
    out = array
    for mask in reversed(mask_list):
        out = inflate_array(out, mask)
    return out


# --------------------------------------------------------------------------
# test: whole-sample driver (the numeric content of toolTest)
# --------------------------------------------------------------------------
def test_sample(sample, sample_binsize, reference, minzscore=None, chromosomes=None,
                mineffectsize=0, multitest=1000, minrefbins=25, repeats=5):
    """Everything toolTest computes between loading and saving (wisecondor.py:174-280).

    `reference` is a mapping with the keys of a reference .npz;
    `sample` the chrom -> int32[] dict of a converted sample.  Returns a dict
    with the arrays toolTest would store.
    """
    if chromosomes is None:
        chromosomes = list(range(1, 23))
    binsize = reference["binsize"].item() if hasattr(reference["binsize"], "item") else reference["binsize"]
    indexes = reference["indexes"]
    distances = reference["distances"]
    chromosome_sizes = [int(v) for v in reference["chromosome_sizes"]]
    mask = reference["mask"]
    masked_sizes = [int(v) for v in reference["masked_sizes"]]
    masked_sums = [sum(masked_sizes[:i + 1]) for i in range(len(masked_sizes))]

    sample = scale_sample(sample, sample_binsize, binsize)
    data = to_numpy_ref_format(sample, chromosome_sizes, mask)
    data = apply_pca(data, reference["pca_mean"], reference["pca_components"])
    cutoff, _ = get_optimal_cutoff(distances, 3)
    thr = z_threshold(masked_sizes, multitest, minzscore)

    z, r, ref_sizes, sd_avg = repeat_test(np.copy(data), indexes, distances, masked_sizes,
                                          masked_sums, cutoff, thr, repeats)
    keep = ref_sizes >= minrefbins
    mask_list = [mask, keep]
    clean_r = r[keep]
    clean_z = z[keep]
    clean_sums = [int(np.sum(keep[:v])) for v in masked_sums]
    clean_bins = [clean_sums[0]] + [clean_sums[i] - clean_sums[i - 1] for i in range(1, len(clean_sums))]

    survivors = inflate_array_multi(np.ones(clean_z.shape, dtype=bool), mask_list)
    calls, chrom_wide = [], []
    for c in [v - 1 for v in chromosomes]:
        lo = sum(clean_bins[:c])
        hi = sum(clean_bins[:c + 1])
        tri = fill_tri_min(clean_z[lo:hi], clean_r[lo:hi], mineffectsize)
        edge = hi - lo
        chrom_wide.append(tri[tri_offset(edge, 0, edge - 1)])
        for value, (x, y) in segment_tri(tri, edge, thr, 3):
            # wisecondor.py:242-253: walk the survivor mask; the end walk restarts
            # at `start` and re-counts it, so end = position(survivor y-1)+1.
            base = sum(chromosome_sizes[:c])
            pos = base
            filled = 0
            while filled <= x:
                filled += survivors[pos] != 0
                pos += 1
            pos -= 1
            end = pos
            while filled <= y:
                filled += survivors[end] != 0
                end += 1
            with np.errstate(**_ERR):
                effect = np.median(clean_r[lo + x:lo + y + 1]) - 1
            calls.append([c + 1, pos - base, end - base, value, effect])

    infl_z = inflate_array_multi(clean_z, mask_list)
    infl_r = inflate_array_multi(clean_r - 1, mask_list)
    res_z, res_r = [], []
    at = 0
    for size in chromosome_sizes:
        res_z.append(infl_z[at:at + size])
        res_r.append(infl_r[at:at + size])
        at += size
    return dict(binsize=binsize, results_z=res_z, results_r=res_r,
                results_cwz=np.array(chrom_wide), results_calls=np.array(calls),
                threshold_z=thr, asdef=sd_avg, aasdef=sd_avg * thr,
                z=z, r=r, ref_sizes=ref_sizes, cutoff=cutoff, data=data)
