"""Sample-file ingest and result-file output for many samples.

The upstream tools read one converted sample after the other with np.load inside the
driver loop (wisecondor.py:75-80, 193-196) and `test` handles one file per process.
Once the GPU side takes a fraction of a millisecond per sample, decoding the .npz
members (zip inflate + unpickle of the chromosome dict) and encoding the result files
is the ceiling -- and in Python both hold the GIL for most of their time.  So both run in
the library's C++ thread pools (csrc/npzio.cpp: zip, a small pickle machine, zlib), batches are
double buffered, and the dense count rows are staged in pinned host memory for the copy
engine.  Nothing numeric happens here.
"""
import argparse
import collections
import concurrent.futures
import ctypes
import os
import pickle
import time

import numpy as np

from . import _lib
from . import wisetools as wt

LoadedSamples = collections.namedtuple('LoadedSamples', 'samples binsizes')


def read_sample(path, to_binsize=None):
    """(chromosome -> int32[bins] dict at `to_binsize`, the file's own bin size)."""
    stored = np.load(path, allow_pickle=True, encoding='latin1')
    own = stored['arguments'].item()['binsize']
    return wt.scaleSample(stored['sample'].item(), own, to_binsize), own


def load_samples(paths, to_binsize=None, threads=8, verbose=False):
    """All samples of `paths`, decoded in a thread pool, in the given order."""
    began = time.time()
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        got = list(pool.map(lambda p: read_sample(p, to_binsize), paths))
    if verbose:
        for path, (_, own) in zip(paths, got):
            print('read %s (binsize %d)' % (path, int(own)))
        print('%d sample files in %.2f s' % (len(paths), time.time() - began))
    return LoadedSamples([g[0] for g in got], set(g[1] for g in got))


def _count_row(path, to_binsize, sizes, out_row):
    sample, _ = read_sample(path, to_binsize)
    out_row[:] = wt.samples_to_counts([sample], sizes)[0]


# ---------------------------------------------------------------- native I/O ----
def _c_strings(items):
    arr = (ctypes.c_char_p * len(items))()
    arr[:] = [os.fsencode(p) for p in items]
    return arr


def read_counts(paths, chrom_sizes, to_binsize, out_rows, threads=8, fallbacks=None):
    """Rows of `out_rows` (int32 [>= len(paths), sum(chrom_sizes)], C-contiguous rows) = the dense
    count vectors of the sample files, decoded by the library's C++ thread pool
    (wc_read_samples: zip inflate, the pickled chromosome dict, scaleSample, pad / truncate).
    A file the native reader does not understand is read with np.load instead -- same result,
    and np.load's own error if the file is broken.  Returns the files' own bin sizes; `fallbacks`
    (a list) collects the positions of the files that took the np.load path."""
    n = len(paths)
    sizes = np.ascontiguousarray(chrom_sizes, dtype=np.int64)
    assert out_rows.dtype == np.int32 and out_rows.shape[1] == int(sizes.sum()) and out_rows.strides[1] == 4
    own = np.zeros(n, dtype=np.float64)
    status = np.zeros(n, dtype=np.int32)
    if n == 0:
        return own
    _lib.check(_lib.load().wc_read_samples(_c_strings(paths), n, int(threads), _lib.ptr(sizes), len(sizes),
                                           float(to_binsize or 0.0), ctypes.c_void_p(out_rows.ctypes.data),
                                           out_rows.strides[0] // 4, _lib.ptr(own), _lib.ptr(status)))
    if fallbacks is not None:
        fallbacks.extend(int(i) for i in np.nonzero(status)[0])      # (tests: which files the native reader passed on)
    for i in np.nonzero(status)[0]:
        sample, size = read_sample(paths[i], to_binsize)      # raises what the reference would report
        out_rows[i, :] = wt.samples_to_counts([sample], sizes)[0]
        own[i] = size
    return own


def load_counts(paths, to_binsize=None, threads=8, verbose=False):
    """`newref`'s sample load as one dense matrix: (counts int32 [files, sum(chrom_bins)], chrom_bins[22],
    the set of the files' own bin sizes).  chrom_bins is the per-chromosome maximum over the samples
    (toNumpyArray, wisetools.py:243-250); two native passes over the files (lengths, then rows), np.load
    for any file the native reader passes on."""
    began = time.time()
    n = len(paths)
    lens = np.zeros((n, 22), dtype=np.int64)
    own = np.zeros(n, dtype=np.float64)
    status = np.zeros(n, dtype=np.int32)
    if n:
        _lib.check(_lib.load().wc_read_sample_lengths(_c_strings(paths), n, int(threads), 22, float(to_binsize or 0.0),
                                                      _lib.ptr(lens), _lib.ptr(own), _lib.ptr(status)))
    for i in np.nonzero(status)[0]:
        sample, size = read_sample(paths[i], to_binsize)
        lens[i] = [len(sample[str(c)]) for c in range(1, 23)]
        own[i] = size
    chrom_bins = [int(v) for v in lens.max(axis=0)] if n else [0] * 22
    counts = np.zeros((n, int(sum(chrom_bins))), dtype=np.int32)
    read_counts(paths, chrom_bins, to_binsize, counts, threads=threads)
    if verbose:
        for path, size in zip(paths, own):
            print('read %s (binsize %d)' % (path, int(size)))
        print('%d sample files in %.2f s' % (n, time.time() - began))
    return counts, chrom_bins, set(float(v) for v in own)


def _object_npy(obj):
    """Bytes of the .npy member np.savez writes for a Python object (a 0-d object array)."""
    header = b"{'descr': '|O', 'fortran_order': False, 'shape': (), }"
    pad = (64 - (10 + len(header) + 1) % 64) % 64
    header = header + b' ' * pad + b'\n'
    body = pickle.dumps(np.array(obj, dtype=object), protocol=3)
    return b'\x93NUMPY\x01\x00' + len(header).to_bytes(2, 'little') + header + body


def write_results(out_paths, per_file_args, runtime, binsize, threshold_z, chrom_sizes, z, r, cwz, calls, n_calls,
                  asdef, threads=8, level=1):
    """One `test` output file per row through the library's C++ thread pool (wc_write_test_results):
    the keys, dtypes and shapes of writeTestOutput's files.  z, r: float64
    [n, sum(chrom_sizes)]; cwz [n, n_sel]; calls [n, max_calls, 5]; n_calls int32 [n]; asdef [n]."""
    n = len(out_paths)
    if n == 0:
        return
    sizes = np.ascontiguousarray(chrom_sizes, dtype=np.int64)
    blobs = [_object_npy(vars(a) if not isinstance(a, dict) else a) for a in per_file_args]
    blob_ptrs = (ctypes.c_char_p * n)()
    blob_ptrs[:] = blobs
    lens = np.array([len(b) for b in blobs], dtype=np.int64)
    rt = _object_npy(runtime)
    status = np.zeros(n, dtype=np.int32)
    for a in (z, r, cwz, calls, asdef):
        assert a.dtype == np.float64 and a.flags['C_CONTIGUOUS']
    assert n_calls.dtype == np.int32 and z.shape[1] == int(sizes.sum())
    _lib.check(_lib.load().wc_write_test_results(
        n, int(threads), _c_strings(out_paths), blob_ptrs, _lib.ptr(lens), rt, len(rt), float(binsize),
        int(isinstance(binsize, (int, np.integer)) and not isinstance(binsize, bool)), float(threshold_z), _lib.ptr(sizes), len(sizes), _lib.ptr(z), _lib.ptr(r), z.shape[1], _lib.ptr(cwz),
        cwz.shape[1], _lib.ptr(calls), _lib.ptr(n_calls), calls.shape[1], _lib.ptr(asdef), int(level), _lib.ptr(status)))
    bad = np.nonzero(status)[0]
    if len(bad):
        raise IOError('could not write %d result file(s), first: %s' % (len(bad), out_paths[bad[0]]))


class _Staging(object):
    """Pinned host buffers of one in-flight batch."""

    def __init__(self, torch, batch, n_total, n_sel, max_calls):
        pin = dict(pin_memory=True)
        self.counts = torch.zeros((batch, n_total), dtype=torch.int32, **pin)
        self.z = torch.empty((batch, n_total), dtype=torch.float64, **pin)
        self.r = torch.empty((batch, n_total), dtype=torch.float64, **pin)
        self.cwz = torch.empty((batch, n_sel), dtype=torch.float64, **pin)
        self.calls = torch.empty((batch, max_calls, 5), dtype=torch.float64, **pin)
        self.n_calls = torch.empty((batch,), dtype=torch.int32, **pin)
        self.asdef = torch.empty((batch,), dtype=torch.float64, **pin)
        self.write_done = None     # future of the writer that still reads these buffers


def output_names(paths, outdir):
    """<outdir>/<leaf>_test.npz per input; two inputs with the same leaf name would overwrite each
    other's result, so that is an error up front."""
    names, seen = [], {}
    for path in paths:
        leaf = os.path.basename(path)
        leaf = leaf[:-4] if leaf.endswith('.npz') else leaf
        if leaf in seen:
            raise ValueError('testbatch: %s and %s would both be written to %s_test.npz; rename one of them'
                             % (seen[leaf], path, leaf))
        seen[leaf] = path
        names.append(os.path.join(outdir, leaf + '_test.npz'))
    return names


def run_testbatch(reference, paths, outdir, threshold, args, writer=None, max_calls=256, runtime=None):
    """`test` for every file of `paths` in GPU batches of args.batch samples.

    Pipeline per batch: decode (C++ thread pool) -> pinned counts -> H2D -> wc_test_batch_dev ->
    D2H into pinned results -> encode + write (C++ thread pool).  The decode of batch i+1 and the
    writes of batch i-1 overlap the GPU work of batch i (two Python helper threads that only wait on
    the native calls, which release the GIL).  Returns timing figures (files, wall_s, files_per_s,
    gpu_s).  `writer(path, per_sample_args, result)`, when given, stores each result file in Python
    instead (the pre-native path, kept for comparison)."""
    import torch
    from .distributed import TestBatch
    began = time.time()
    n = len(paths)
    if n == 0:
        return dict(files=0, wall_s=0.0, files_per_s=0.0, gpu_s=0.0)
    dev = torch.device('cuda', reference.device)
    torch.cuda.set_device(dev)
    sizes = [int(v) for v in reference.chromosome_sizes]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    sel = list(args.chromosomes)
    # wc_test_batch_dev takes at most 60000 (sample, chromosome) regions per call
    batch = max(1, min(int(args.batch), n, 60000 // max(1, len(sel))))
    io_threads = max(1, int(getattr(args, 'io', 8)))
    level = int(getattr(args, 'ziplevel', 1))
    outs = output_names(paths, outdir)
    stage = [_Staging(torch, batch, reference.n_total, len(sel), max_calls) for _ in range(2)]
    dev_counts = torch.zeros((batch, reference.n_total), dtype=torch.int32, device=dev)

    def new_batch(counts, calls_cap):
        return TestBatch(reference, counts, threshold, minrefbins=args.minrefbins, repeats=args.repeats,
                         chromosomes=sel, max_calls=calls_cap, mineffectsize=args.mineffectsize)
    tb = new_batch(dev_counts, max_calls)
    helpers = concurrent.futures.ThreadPoolExecutor(max_workers=2)
    py_writers = concurrent.futures.ThreadPoolExecutor(max_workers=io_threads) if writer else None
    gpu_s = wait_s = run_s = d2h_s = 0.0      # the GPU section and its parts: waiting for the result writers' buffers,
                                              # H2D + wc_test_batch_dev, D2H of the results
    pending = []

    def start_decode(at, slot):
        names = paths[at:at + batch]
        rows = stage[slot].counts.numpy()               # (the result writers never touch the counts buffer)
        return at, names, helpers.submit(read_counts, names, sizes, reference.binsize, rows, io_threads)

    def emit(at, names, slot, ns):
        st = stage[slot]
        per_file = []
        for i, name in enumerate(names):
            # what one `test` call would record: its own infile / outfile, not the batch's whole file list
            one = argparse.Namespace(**{k: v for k, v in vars(args).items() if k not in ('infiles', 'func')})
            one.infile = name
            one.outfile = outs[at + i]
            per_file.append(one)
        if writer is None:
            st.write_done = helpers.submit(
                write_results, outs[at:at + ns], per_file, runtime, reference.binsize, threshold, sizes,
                st.z.numpy()[:ns], st.r.numpy()[:ns], st.cwz.numpy()[:ns], st.calls.numpy()[:ns],
                st.n_calls.numpy()[:ns], st.asdef.numpy()[:ns], io_threads, level)
            pending.append(st.write_done)
            return
        z, r = st.z.numpy(), st.r.numpy()
        cwz, calls, n_calls, asdef = st.cwz.numpy(), st.calls.numpy(), st.n_calls.numpy(), st.asdef.numpy()
        for i, one in enumerate(per_file):
            result = dict(
                results_z=[z[i, offs[c]:offs[c + 1]].copy() for c in range(len(sizes))],
                results_r=[r[i, offs[c]:offs[c + 1]].copy() for c in range(len(sizes))],
                results_cwz=cwz[i].copy(), results_calls=calls[i, :n_calls[i]].copy(), asdef=float(asdef[i]))
            pending.append(py_writers.submit(writer, one.outfile, one, result))

    try:
        nxt = start_decode(0, 0)
        for bi, at in enumerate(range(0, n, batch)):
            slot = bi & 1
            at_, names, fut = nxt
            fut.result()
            if at + batch < n:
                nxt = start_decode(at + batch, slot ^ 1)
            ns = len(names)
            t0 = time.time()
            st = stage[slot]
            if st.write_done is not None:
                st.write_done.result()                  # the writer of two batches ago is done with these buffers
                st.write_done = None
            wait_s += time.time() - t0
            t1 = time.time()
            while True:
                dev_counts[:ns].copy_(st.counts[:ns], non_blocking=True)
                run = tb if ns == batch else new_batch(dev_counts[:ns], tb.max_calls)
                try:
                    run.run()
                except _lib.WisecondorHipError as exc:
                    # a sample with more calls than the output holds (e.g. hardly any reads):
                    # the reference has no such limit, so run the batch again with more room
                    if getattr(exc, 'code', 0) == _lib.E_LIMIT and 'max_calls' in str(exc) \
                            and tb.max_calls < reference.n_total:
                        grown = tb.max_calls * 4
                        tb = new_batch(dev_counts, grown)
                        for s_ in (0, 1):
                            if stage[s_].write_done is not None:
                                stage[s_].write_done.result()
                                stage[s_].write_done = None
                            stage[s_].calls = torch.empty((batch, grown, 5), dtype=torch.float64, pin_memory=True)
                        continue
                    raise
                break
            torch.cuda.synchronize()
            run_s += time.time() - t1
            t2 = time.time()
            st.z[:ns].copy_(run.results_z, non_blocking=True)
            st.r[:ns].copy_(run.results_r, non_blocking=True)
            st.cwz[:ns].copy_(run.cwz, non_blocking=True)
            st.calls[:ns].copy_(run.calls, non_blocking=True)
            st.n_calls[:ns].copy_(run.n_calls, non_blocking=True)
            st.asdef[:ns].copy_(run.asdef, non_blocking=True)
            torch.cuda.synchronize()
            d2h_s += time.time() - t2
            gpu_s += time.time() - t0
            emit(at, names, slot, ns)
        for f in pending:
            f.result()
    finally:
        helpers.shutdown(wait=True)
        if py_writers:
            py_writers.shutdown(wait=True)
    wall = time.time() - began
    return dict(files=n, wall_s=wall, files_per_s=n / wall if wall > 0 else 0.0, gpu_s=gpu_s,
                wait_writers_s=wait_s, h2d_and_kernels_s=run_s, d2h_s=d2h_s)
