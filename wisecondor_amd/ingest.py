"""Sample-file ingest and result-file output for many samples.

The upstream tools read one converted sample after the other with np.load inside the
driver loop (wisecondor.py:75-80, 193-196) and `test` handles one file per process.
Once the GPU side takes a fraction of a millisecond per sample, decoding the .npz
members (zip inflate + unpickle of the chromosome dict) and encoding the result files
is the ceiling, so both run in thread pools (zlib releases the GIL), batches are
double buffered, and the dense count rows are staged in pinned host memory for the
copy engine.  Nothing numeric happens here.
"""
import argparse
import collections
import concurrent.futures
import os
import time

import numpy as np

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


def run_testbatch(reference, paths, outdir, threshold, args, writer, max_calls=256):
    """`test` for every file of `paths` in GPU batches of args.batch samples.

    Pipeline per batch: decode (thread pool) -> pinned counts -> H2D -> wc_test_batch_dev ->
    D2H into pinned results -> encode + write (thread pool).  Decode of batch i+1 and the
    writes of batch i-1 overlap the GPU work of batch i.  Returns timing figures
    (files, wall_s, files_per_s, gpu_s).  `writer(path, per_sample_args, result)` stores one
    result file."""
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
    batch = max(1, min(int(args.batch), n))
    io_threads = max(1, int(getattr(args, 'io', 8)))
    stage = [_Staging(torch, batch, reference.n_total, len(sel), max_calls) for _ in range(2)]
    dev_counts = torch.zeros((batch, reference.n_total), dtype=torch.int32, device=dev)
    tb = TestBatch(reference, dev_counts, threshold, minrefbins=args.minrefbins, repeats=args.repeats,
                   chromosomes=sel, max_calls=max_calls, mineffectsize=args.mineffectsize)
    readers = concurrent.futures.ThreadPoolExecutor(max_workers=io_threads)
    writers = concurrent.futures.ThreadPoolExecutor(max_workers=io_threads)
    gpu_s = 0.0
    pending_writes = []

    def start_decode(at, slot):
        names = paths[at:at + batch]
        rows = stage[slot].counts.numpy()
        return names, [readers.submit(_count_row, name, reference.binsize, sizes, rows[i])
                       for i, name in enumerate(names)]

    def emit(names, slot):
        st = stage[slot]
        z, r = st.z.numpy(), st.r.numpy()
        cwz, calls, n_calls, asdef = st.cwz.numpy(), st.calls.numpy(), st.n_calls.numpy(), st.asdef.numpy()
        for i, name in enumerate(names):
            result = dict(
                results_z=[z[i, offs[c]:offs[c + 1]].copy() for c in range(len(sizes))],
                results_r=[r[i, offs[c]:offs[c + 1]].copy() for c in range(len(sizes))],
                results_cwz=cwz[i].copy(), results_calls=calls[i, :n_calls[i]].copy(), asdef=float(asdef[i]))
            leaf = os.path.basename(name)
            leaf = leaf[:-4] if leaf.endswith('.npz') else leaf
            one = argparse.Namespace(**vars(args))
            one.infile = name
            one.outfile = os.path.join(outdir, leaf + '_test.npz')
            pending_writes.append(writers.submit(writer, one.outfile, one, result))

    try:
        nxt = start_decode(0, 0)
        for bi, at in enumerate(range(0, n, batch)):
            slot = bi & 1
            names, futs = nxt
            for f in futs:
                f.result()
            if at + batch < n:
                nxt = start_decode(at + batch, slot ^ 1)
            ns = len(names)
            t0 = time.time()
            while True:
                dev_counts[:ns].copy_(stage[slot].counts[:ns], non_blocking=True)
                run = tb if ns == batch else TestBatch(reference, dev_counts[:ns], threshold,
                                                      minrefbins=args.minrefbins, repeats=args.repeats,
                                                      chromosomes=sel, max_calls=tb.max_calls,
                                                      mineffectsize=args.mineffectsize)
                try:
                    run.run()
                except Exception as exc:
                    # a sample with more calls than the output holds (e.g. hardly any reads):
                    # the reference has no such limit, so run the batch again with more room
                    if 'max_calls' in str(exc) and tb.max_calls < reference.n_total:
                        grown = tb.max_calls * 4
                        tb = TestBatch(reference, dev_counts, threshold, minrefbins=args.minrefbins,
                                       repeats=args.repeats, chromosomes=sel, max_calls=grown,
                                       mineffectsize=args.mineffectsize)
                        for s in (0, 1):
                            stage[s].calls = torch.empty((batch, grown, 5), dtype=torch.float64, pin_memory=True)
                        continue
                    raise
                break
            st = stage[slot]
            st.z[:ns].copy_(run.results_z, non_blocking=True)
            st.r[:ns].copy_(run.results_r, non_blocking=True)
            st.cwz[:ns].copy_(run.cwz, non_blocking=True)
            st.calls[:ns].copy_(run.calls, non_blocking=True)
            st.n_calls[:ns].copy_(run.n_calls, non_blocking=True)
            st.asdef[:ns].copy_(run.asdef, non_blocking=True)
            torch.cuda.synchronize()
            gpu_s += time.time() - t0
            # this slot's host buffers are rewritten two batches from now: the writers of that
            # earlier batch must be done with their (copied) slices before then -- they copy above
            emit(names, slot)
        for f in pending_writes:
            f.result()
    finally:
        readers.shutdown(wait=True)
        writers.shutdown(wait=True)
    wall = time.time() - began
    return dict(files=n, wall_s=wall, files_per_s=n / wall if wall > 0 else 0.0, gpu_s=gpu_s)
