"""The native sample reader / result writer (csrc/npzio.cpp) against numpy's
own np.load / np.savez on the same files.  Host only: runs without a GPU."""
import argparse
import io
import os
import pickle
import struct
import zipfile

import numpy as np
import pytest

from wisecondor_amd import ingest
from wisecondor_amd import wisetools as wt

KEYS = [str(c) for c in range(1, 23)] + ["X", "Y"]


def _sample(seed, lengths, dtype=np.int32):
    rng = np.random.RandomState(seed)
    return {k: rng.poisson(40.0, size=n).astype(dtype) for k, n in zip(KEYS, lengths)}


def _npy_object(obj, protocol):
    buf = io.BytesIO()
    np.lib.format.write_array_header_1_0(buf, {"descr": "|O", "fortran_order": False, "shape": ()})
    return buf.getvalue() + pickle.dumps(np.array(obj, dtype=object), protocol=protocol)


def _write_with_protocol(path, sample, binsize, protocol, compress=True):
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED if compress else zipfile.ZIP_STORED) as z:
        z.writestr("arguments.npy", _npy_object({"binsize": binsize, "infile": "x.bam", "retdist": 4}, protocol))
        z.writestr("runtime.npy", _npy_object({}, protocol))
        z.writestr("sample.npy", _npy_object(sample, protocol))
        z.writestr("quality.npy", _npy_object({"mapped": 1}, protocol))


def _py2_pickle_of_sample(sample, binsize=None):
    """The opcode stream Python 2's numpy wrote for `np.array(dict, dtype=object)`: protocol 2, GLOBAL
    references, str (not bytes) payloads in BINSTRING / SHORT_BINSTRING -- what the reference's own
    `convert` leaves on disk."""
    def s_(b):
        return (b"U" + bytes([len(b)]) + b) if len(b) < 256 else (b"T" + struct.pack("<i", len(b)) + b)

    def int_(v):
        return b"K" + bytes([v]) if 0 <= v < 256 else (b"M" + struct.pack("<H", v) if v < 65536 else b"J" + struct.pack("<i", v))

    def dtype_(code, endian, flags):
        return (b"cnumpy\ndtype\n" + s_(code) + b"K\x00K\x01\x87R(K\x03" + s_(endian) + b"NNNJ\xff\xff\xff\xffJ\xff\xff\xff\xff"
                + int_(flags) + b"tb")
    recon = b"cnumpy.core.multiarray\n_reconstruct\ncnumpy\nndarray\nK\x00\x85" + s_(b"b") + b"\x87R"
    out = b"\x80\x02" + recon + b"(K\x01)" + dtype_(b"O8", b"|", 63) + b"\x89]}("
    if binsize is not None:
        out += s_(b"binsize") + b"G" + struct.pack(">d", binsize)
    else:
        for k, v in sample.items():
            v = np.ascontiguousarray(v, dtype="<i4")
            out += s_(k.encode()) + recon + b"(K\x01" + int_(len(v)) + b"\x85" + dtype_(b"i4", b"<", 0) + b"\x89" + s_(v.tobytes()) + b"tb"
    out += b"uatb."
    buf = io.BytesIO()
    np.lib.format.write_array_header_1_0(buf, {"descr": "|O", "fortran_order": False, "shape": ()})
    return buf.getvalue() + out


def _reference_rows(paths, sizes, to_binsize):
    rows = np.zeros((len(paths), int(np.sum(sizes))), dtype=np.int32)
    for i, p in enumerate(paths):
        ingest._count_row(p, to_binsize, sizes, rows[i])
    return rows


def test_reader_matches_np_load(tmp_path):
    sizes = [30 + 2 * c for c in range(22)]
    paths = []
    # ragged files: some chromosomes longer (truncate), some shorter (zero pad) than the reference's
    for i in range(6):
        lengths = [max(1, n + (i - 3) * (c % 3)) for c, n in enumerate(sizes)] + [17, 9]
        p = str(tmp_path / ("s%d.npz" % i))
        np.savez_compressed(p, arguments={"binsize": 250000.0, "infile": "a.bam"}, runtime={},
                            sample=_sample(i, lengths, np.int64 if i == 4 else np.int32), quality={})
        paths.append(p)
    # older pickle protocols, stored (not deflated) members, integer bin size
    for proto in (2, 3):
        p = str(tmp_path / ("p%d.npz" % proto))
        _write_with_protocol(p, _sample(10 + proto, sizes + [5, 5]), 250000 if proto == 2 else 250000.0, proto,
                             compress=proto == 3)
        paths.append(p)
    want = _reference_rows(paths, sizes, 250000.0)
    got = np.full_like(want, -7)
    slow = []
    own = ingest.read_counts(paths, sizes, 250000.0, got, threads=3, fallbacks=slow)
    assert slow == []                                   # every one of these went through the native reader
    assert np.array_equal(got, want)
    assert np.all(own == 250000.0)


def test_reader_python2_files_and_scaling(tmp_path):
    sizes = [12 + c for c in range(22)]
    fine = [5 * n - (c % 4) for c, n in enumerate(sizes)] + [40, 11]        # 50 kb bins, ragged ends
    sample = _sample(3, fine)
    p2 = str(tmp_path / "py2.npz")
    with zipfile.ZipFile(p2, "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("sample.npy", _py2_pickle_of_sample(sample))
        z.writestr("arguments.npy", _py2_pickle_of_sample(None, binsize=50000.0))
    back = np.load(p2, allow_pickle=True, encoding="latin1")
    assert back["arguments"].item()["binsize"] == 50000.0 and np.array_equal(back["sample"].item()["7"], sample["7"])
    p3 = str(tmp_path / "py3.npz")
    np.savez_compressed(p3, arguments={"binsize": 50000.0}, runtime={}, sample=sample, quality={})
    want = _reference_rows([p2, p3], sizes, 250000.0)
    assert want.sum() > 0 and np.array_equal(want[0], want[1])
    got = np.zeros_like(want)
    slow = []
    own = ingest.read_counts([p2, p3], sizes, 250000.0, got, threads=2, fallbacks=slow)
    assert slow == []
    assert np.array_equal(got, want) and list(own) == [50000.0, 50000.0]
    # no scaling asked: the fine bins, padded / truncated
    want = _reference_rows([p2], sizes, None)
    got = np.zeros_like(want)
    ingest.read_counts([p2], sizes, None, got)
    assert np.array_equal(got, want)


def test_reader_falls_back_and_reports(tmp_path):
    sizes = [10] * 22
    odd = {k: np.arange(10, dtype=np.float16) for k in KEYS}               # a dtype the native reader does not take
    p = str(tmp_path / "odd.npz")
    np.savez_compressed(p, arguments={"binsize": 1e6}, runtime={}, sample=odd, quality={})
    got = np.zeros((1, 220), dtype=np.int32)
    slow = []
    ingest.read_counts([p], sizes, 1e6, got, fallbacks=slow)
    assert slow == [0]
    assert np.array_equal(got[0, :10], np.arange(10))
    # an impossible rescale is the reference's ERROR + exit(1) (wisetools.py:224-226), worded by the Python path
    with pytest.raises(SystemExit):
        ingest.read_counts([p], sizes, 1500000.0, got)
    with pytest.raises(Exception):
        ingest.read_counts([str(tmp_path / "missing.npz")], sizes, 1e6, got)


def test_reader_rejects_damaged_zip_headers_without_reading_past_the_file(tmp_path):
    """Sizes and offsets of a zip member come from the file.  Damaged ones -- values that would wrap a size_t in
    `offset + size` checks, a zip64 extra field claiming 2^63 bytes, a local header beyond the end, a member that
    claims to inflate to a terabyte -- must make the native reader pass the file on (status != 0) instead of reading
    past the buffer or allocating what the header says; np.load then words the error."""
    import struct
    from wisecondor_amd import _lib
    sizes = [10] * 22
    good = str(tmp_path / "good.npz")
    np.savez_compressed(good, arguments={"binsize": 1e6}, runtime={}, sample=_sample(3, [10] * 24), quality={})
    raw = bytearray(open(good, "rb").read())
    eocd = raw.rfind(b"PK\x05\x06")
    cd_off = struct.unpack_from("<I", raw, eocd + 16)[0]
    n_entries = struct.unpack_from("<H", raw, eocd + 10)[0]
    # central directory entry of sample.npy
    at, entry = cd_off, None
    for _ in range(n_entries):
        nl, xl, cl = struct.unpack_from("<HHH", raw, at + 28)
        if bytes(raw[at + 46:at + 46 + nl]) == b"sample.npy":
            entry = at
        at += 46 + nl + xl + cl
    assert entry is not None

    def variant(name, edit):
        data = bytearray(raw)
        edit(data)
        path = str(tmp_path / name)
        open(path, "wb").write(bytes(data))
        return path

    def huge_usize(d):       # inflates to 2^40 bytes, says the header
        struct.pack_into("<I", d, entry + 24, 0xFFFFFFFE)

    def huge_csize(d):       # compressed size beyond the file: offset + size wraps in a careless check
        struct.pack_into("<I", d, entry + 20, 0xFFFFFFF0)

    def far_local(d):        # local header offset beyond the end of the file
        struct.pack_into("<I", d, entry + 42, 0xFFFFFF00)

    def far_directory(d):    # central directory offset beyond the end of the file
        struct.pack_into("<I", d, eocd + 16, 0xFFFFFF00)

    def many_entries(d):     # more directory entries than the file could hold
        struct.pack_into("<H", d, eocd + 10, 0xFFFE)
        struct.pack_into("<H", d, eocd + 8, 0xFFFE)

    paths = [variant("v%d.npz" % i, e) for i, e in enumerate((huge_usize, huge_csize, far_local, far_directory, many_entries))]
    # zip64 extra field with 2^63-sized members, appended to the directory entry of a copy
    data = bytearray(raw)
    nl, xl, cl = struct.unpack_from("<HHH", data, entry + 28)
    extra = struct.pack("<HHQQ", 1, 16, 1 << 63, 1 << 63)
    struct.pack_into("<I", data, entry + 20, 0xFFFFFFFF)
    struct.pack_into("<I", data, entry + 24, 0xFFFFFFFF)
    struct.pack_into("<H", data, entry + 30, xl + len(extra))
    data[entry + 46 + nl + xl:entry + 46 + nl + xl] = extra
    new_eocd = eocd + len(extra)
    struct.pack_into("<I", data, new_eocd + 12, struct.unpack_from("<I", data, new_eocd + 12)[0] + len(extra))
    p64 = str(tmp_path / "zip64.npz")
    open(p64, "wb").write(bytes(data))
    paths.append(p64)
    paths.append(good)
    n = len(paths)
    rows = np.zeros((n, 220), dtype=np.int32)
    own = np.zeros(n)
    status = np.zeros(n, dtype=np.int32)
    csz = np.ascontiguousarray(sizes, dtype=np.int64)
    _lib.check(_lib.load().wc_read_samples(ingest._c_strings(paths), n, 2, _lib.ptr(csz), 22, 1e6, _lib.ptr(rows), 220,
                                           _lib.ptr(own), _lib.ptr(status)))
    assert status[-1] == 0 and rows[-1].sum() > 0           # the untouched file reads fine
    assert (status[:-1] != 0).all(), status                  # every damaged one is passed on, nothing crashed
    # the np.load path either words an error or -- where Python's zipfile trusts a different copy of the damaged
    # field -- still delivers the file's true content; never something else
    for path in paths[:-1]:
        got = np.zeros((1, 220), dtype=np.int32)
        try:
            ingest.read_counts([path], sizes, 1e6, got)
        except Exception:
            continue
        assert np.array_equal(got[0], rows[-1]), path


def test_writer_matches_np_savez(tmp_path):
    from wisecondor_amd import wisecondor as cli
    rng = np.random.RandomState(5)
    sizes = [7 + (c % 5) for c in range(22)]
    n_total, n = int(np.sum(sizes)), 4
    offs = np.concatenate([[0], np.cumsum(sizes)])
    z = rng.standard_normal((n, n_total))
    z[:, ::7] = 0.0
    r = rng.standard_normal((n, n_total)) * 0.01
    cwz = rng.standard_normal((n, 22))
    calls = rng.standard_normal((n, 8, 5))
    n_calls = np.array([3, 0, 8, 1], dtype=np.int32)
    asdef = rng.uniform(0.9, 1.1, size=n)
    args = cli.buildParser().parse_args(["test", "in.npz", "out.npz", "ref.npz"])
    per_file, outs = [], []
    for i in range(n):
        one = argparse.Namespace(**vars(args))
        one.infile, one.outfile = "in_%d.npz" % i, str(tmp_path / ("native_%d.npz" % i))
        per_file.append(one)
        outs.append(one.outfile)
    runtime = {"version": "abc", "datetime": "now", "hostname": "h", "username": "u"}
    # binsize as the reference file holds it: a float (the reference's CLI, type=float) or an int (this CLI's
    # -binsize): the Python value goes into the file unchanged, so the member's dtype follows it
    for level, binsize in ((1, 250000.0), (0, 250000)):
        ingest.write_results(outs, per_file, runtime, binsize, 5.25, sizes, z, r, cwz, calls, n_calls, asdef,
                             threads=3, level=level)
        for i in range(n):
            result = dict(results_z=[z[i, offs[c]:offs[c + 1]] for c in range(22)],
                          results_r=[r[i, offs[c]:offs[c + 1]] for c in range(22)],
                          results_cwz=cwz[i], results_calls=calls[i, :n_calls[i]], asdef=float(asdef[i]))
            ref_path = str(tmp_path / ("python_%d.npz" % i))
            cli.writeTestOutput(ref_path, per_file[i], binsize, result, 5.25)
            a = np.load(ref_path, allow_pickle=True)
            b = np.load(outs[i], allow_pickle=True)
            assert sorted(a.files) == sorted(b.files)
            for key in a.files:
                x, y = a[key], b[key]
                if key == "runtime":
                    assert y.item() == runtime
                    continue
                assert x.dtype == y.dtype and x.shape == y.shape, key
                if x.dtype == object and x.shape == ():
                    assert x.item() == y.item(), key
                elif x.dtype == object:
                    assert all(p.dtype == q.dtype and np.array_equal(p, q) for p, q in zip(x, y)), key
                else:
                    assert np.array_equal(x, y), key
            assert b["binsize"].dtype == (np.int64 if isinstance(binsize, int) else np.float64)
            assert b["results_calls"].shape == ((n_calls[i], 5) if n_calls[i] else (0,))
            with zipfile.ZipFile(outs[i]) as zf:
                assert zf.testzip() is None
                assert {m.compress_type for m in zf.infolist() if m.file_size > 64} == \
                    {zipfile.ZIP_DEFLATED if level else zipfile.ZIP_STORED}


def test_parallel_savez_reads_back_like_numpys(tmp_path):
    """wisecondor_amd.npzfast.savez: the .npz container with members deflated chunk-wise in threads
    (sync-flushed chunks concatenated into one raw deflate stream) must read back through np.load
    exactly like np.savez_compressed's file -- values, dtypes, shapes, memory order, pickled objects."""
    from wisecondor_amd import npzfast
    rng = np.random.RandomState(3)
    big = rng.standard_normal((1100, 1000))                    # 8.8 MB: three chunks
    fort = np.asfortranarray(rng.standard_normal((700, 90)))
    arrays = dict(arguments={"infiles": ["a", "b"], "binsize": None, "refsize": 100}, runtime={"version": b"x"},
                  binsize=250000.0, indexes=rng.randint(-1, 5000, size=(3000, 100)).astype(np.int32), distances=big,
                  correctedData=fort, mask=rng.rand(5000) > 0.1, sums=[1, 2, 3], empty=np.zeros((0, 5)),
                  ragged=np.array([np.arange(3.0), np.arange(5.0)], dtype=object))
    ref = str(tmp_path / "numpy.npz")
    np.savez_compressed(ref, **arrays)
    want = np.load(ref, allow_pickle=True)
    for level, threads in ((6, 4), (1, 1), (0, 2)):
        path = npzfast.savez(str(tmp_path / ("fast_%d" % level)), level=level, threads=threads, **arrays)
        assert path.endswith(".npz")
        with zipfile.ZipFile(path) as zf:
            assert zf.testzip() is None
            assert [m.filename for m in zf.infolist()] == [k + ".npy" for k in arrays]
        got = np.load(path, allow_pickle=True)
        assert sorted(got.files) == sorted(want.files)
        for key in want.files:
            a, b = want[key], got[key]
            assert a.dtype == b.dtype and a.shape == b.shape, key
            assert a.flags["F_CONTIGUOUS"] == b.flags["F_CONTIGUOUS"] and a.flags["C_CONTIGUOUS"] == b.flags["C_CONTIGUOUS"], key
            if a.dtype == object and a.shape == ():
                assert a.item() == b.item(), key
            elif a.dtype == object:
                assert all(np.array_equal(p, q) for p, q in zip(a, b)), key
            else:
                assert np.array_equal(a, b), key
    assert os.path.getsize(str(tmp_path / "fast_6.npz")) < 1.02 * os.path.getsize(ref)


def test_load_counts_equals_the_python_sample_load(tmp_path):
    """ingest.load_counts (`newref`'s sample load: native lengths pass + native rows pass) against
    load_samples + samples_to_counts, ragged chromosome lengths, merged bins, one file the native reader
    passes on."""
    sizes = [20 + c for c in range(22)]
    paths = []
    for i in range(5):
        lengths = [5 * n - (i % 3) * (c % 2) for c, n in enumerate(sizes)] + [33, 8]
        p = str(tmp_path / ("s%d.npz" % i))
        np.savez_compressed(p, arguments={"binsize": 50000.0}, runtime={}, sample=_sample(40 + i, lengths), quality={})
        paths.append(p)
    odd = {k: v.astype(np.float32) for k, v in _sample(99, [5 * n for n in sizes] + [5, 5]).items()}
    odd["3"] = odd["3"].astype(np.float16)                                       # -> WC_NPZ_UNSUPPORTED -> np.load
    p = str(tmp_path / "odd.npz")
    np.savez_compressed(p, arguments={"binsize": 50000.0}, runtime={}, sample=odd, quality={})
    paths.append(p)
    for to_binsize in (250000.0, None):
        loaded = ingest.load_samples(paths, to_binsize, threads=2)
        chrom_bins = [max(len(s[str(c)]) for s in loaded.samples) for c in range(1, 23)]
        want = wt.samples_to_counts(loaded.samples, chrom_bins)
        counts, bins, binsizes = ingest.load_counts(paths, to_binsize, threads=3)
        assert bins == chrom_bins and binsizes == {50000.0}
        assert np.array_equal(counts, want)
