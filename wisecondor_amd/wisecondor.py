"""Command line of the MI355X build: the `newref*` and `test` sub-commands of WISECONDOR.

Contract (taken from the upstream CLI at
wisecondor.py:345-521): same sub-command names, positional arguments, single-dash
options, defaults and `dest` names; same file naming (`<stem>_prep.npz`,
`<stem>_part_<m>.npz`, resume by file existence, temporaries removed after a
successful merge); same `.npz` keys and dtypes; `test` ends with exit status 0.
Everything numeric runs on the GPU through libwisecondor_hip.so (no CPU path).

Build-only additions, all off by default: `-gpus N` on `newref` / `testbatch`
(one process per GPU), the `testbatch` sub-command (many samples per GPU batch).
`convert`, `plot` and `report` are not part of this build.
"""
import argparse
import copy
import datetime
import getpass
import os
import socket
import subprocess
import sys
import time

import numpy as np
from scipy.stats import norm

from . import wisetools as wt

_STARTED = datetime.datetime.now()


# ------------------------------------------------------------------ provenance ----
_VERSION_TAG = []


def getRuntime():
    """The `runtime` member of every output file: version, datetime, hostname, username
    (upstream wisetools.py:47-62).  `git describe` runs once per process."""
    if not _VERSION_TAG:
        try:
            _VERSION_TAG.append(subprocess.check_output(["git", "describe", "--always"], stderr=subprocess.DEVNULL).split()[0])
        except Exception:
            _VERSION_TAG.append('unknown')
    tag = _VERSION_TAG[0]
    return {'version': tag, 'datetime': _STARTED, 'hostname': socket.gethostname(),
            'username': getpass.getuser()}


def printArgs(args):
    """Echo the parsed command (tool name first, then the options in name order)."""
    settings = dict(vars(args))
    tool = settings.pop('func', None)
    print('tool =', getattr(tool, '__name__', str(tool)).replace('tool', '', 1))
    for name in sorted(settings):
        print(name, '=', settings[name])


def _open_npz(path):
    return np.load(path, allow_pickle=True, encoding='latin1')


def _as_object_array(arrays):
    """Ragged per-chromosome arrays as one object array (numpy >= 1.24 no longer infers it)."""
    boxed = np.empty(len(arrays), dtype=object)
    for at, item in enumerate(arrays):
        boxed[at] = item
    return boxed


# --------------------------------------------------------------- newref: files ----
class BuildFiles(object):
    """Names of the intermediate files of one `newref` job.

    `reference.npz` -> `reference_prep.npz` and `reference_part_<m>.npz` (a trailing `.npz`
    of the output name is dropped first; upstream wisecondor.py:31-39).
    """

    def __init__(self, outfile):
        folder, leaf = os.path.split(outfile)
        if leaf.endswith('.npz'):
            leaf = leaf[:-len('.npz')]
        stem = os.path.join(folder, leaf)
        self.prep = stem + '_prep.npz'
        self.part_base = stem + '_part'

    @staticmethod
    def part_name(part_base, number):
        return '%s_%d.npz' % (part_base, number)


def _missing_parts(part_base, parts):
    return [m for m in range(1, parts + 1) if not os.path.isfile(BuildFiles.part_name(part_base, m))]


def _visible_gpus():
    import torch
    return int(torch.cuda.device_count())      # counting devices does not initialise the runtime


def _rank_count(args):
    """Processes (= GPUs) a `newref` / `testbatch` run uses.

    `-gpus N` asks for N; without it `-cpus N` (the upstream worker-pool size) is honoured up to
    the number of visible GPUs.  One rank runs in this process, more are started as children
    (wisecondor_amd.ranks) before anything here touches the GPU."""
    want = getattr(args, 'gpus', None)
    if want is None:
        want = min(max(1, int(getattr(args, 'cpus', 1))), max(1, _visible_gpus()))
    if want < 1:
        print('ERROR: -gpus needs a positive count, got', want)
        sys.exit(1)
    return int(want)


def write_npz(path, temporary=False, **arrays):
    """The .npz files of this tool (np.load reads them like np.savez_compressed's, which the
    reference uses: wisecondor.py:97-108, 128-132, 160-170).  Once the GPU part takes
    milliseconds the sub-commands' wall time is zlib (1.7 of 1.8 s of `newref` at 100 x 250 kb),
    so members are deflated at level 1 in threads (wisecondor_amd/npzfast.py; WC_NPZ_LEVEL=6 for
    numpy's own level) and the prep / part files `newref` deletes again at its end are stored."""
    from . import npzfast
    level = 0 if temporary else int(os.environ.get('WC_NPZ_LEVEL', '1'))
    npzfast.savez(path, level=level, threads=min(16, os.cpu_count() or 1), **arrays)


def save_part(part_base, number, parts, indexes, distances, args, temporary=False):
    """One part file exactly as `newrefpart` writes it (keys of SURVEY.md App. B)."""
    stamped = copy.copy(args)
    stamped.part = [number, parts]
    write_npz(BuildFiles.part_name(part_base, number), temporary,
              arguments=vars(stamped),
              runtime=getRuntime(),
              indexes=indexes,
              distances=distances)


def part_owners(parts, numbers, n_bins, world):
    """Which rank owns each of the part files `numbers` (1-based) of `parts`: the rank whose row range
    (distributed.row_range = the reference's getPart for `world` parts) holds the part's rows; None when
    some part straddles two ranks' ranges (then every rank needs all rows)."""
    from .distributed import row_range
    ranges = [row_range(r, world, n_bins) for r in range(world)]
    owners = {}
    for m in numbers:
        lo, hi = wt.getPart(m - 1, parts, n_bins)
        own = [r for r, (b, e) in enumerate(ranges) if b <= lo and hi <= e]
        if hi <= lo:
            own = [r for r, (b, e) in enumerate(ranges) if b <= lo <= e] or [0]
        if not own:
            return None
        owners[m] = own[0]
    return owners


def select_all_rows(prepfile, refsize, device=0, rank=0, world=1, gather=True):
    """indexes / distances of every bin from a prep file, matrix resident on `device`.

    One NewrefJob pass (wisecondor_amd.distributed); with world > 1 the ranks share the work
    and each ends with the full result -- or, gather=False, with the rows it owns only (no result
    all-gather: the reference's workers exchange nothing but files, wisecondor.py:47-56)."""
    import torch
    from . import _lib
    from .distributed import NewrefJob
    prep = prepfile if isinstance(prepfile, dict) else _open_npz(prepfile)
    corrected = np.asarray(prep['correctedData'])
    bins = np.ascontiguousarray(prep['maskedChromBins'], dtype=np.int64)
    order = wt.sum_order_of(corrected)
    dev = torch.device('cuda', device)
    torch.cuda.set_device(dev)
    X = torch.from_numpy(np.ascontiguousarray(corrected, dtype=np.float64)).to(dev)
    job = NewrefJob(_lib.context(device), X, bins, int(refsize), order, rank=rank, world=world, gather=gather)
    idx, dst = job.run()
    torch.cuda.synchronize()
    return idx.cpu().numpy(), dst.cpu().numpy(), job


# --------------------------------------------------------------- newref: tools ----
def toolNewrefPrep(args, temporary=False):
    """`newrefprep`: sample files -> prep file (normalise, mask all-zero bins, PCA-correct).

    File contract of upstream wisecondor.py:72-108; the arithmetic is wt.prepReference (GPU).
    Returns what it wrote (for `newref`, which goes on with the arrays instead of reading the
    file back)."""
    from . import ingest
    counts, chrom_bins_in, binsizes = ingest.load_counts(args.infiles, args.binsize, threads=min(16, os.cpu_count() or 1),
                                                         verbose=True)
    if args.binsize is None and len(binsizes) > 1:
        print('ERROR: the input samples were binned at different sizes:', sorted(binsizes))
        print('Drop the odd ones or give -binsize to merge them to a common size')
        sys.exit(1)
    binsize = args.binsize if args.binsize is not None else next(iter(binsizes))

    n_given = counts.shape[0]
    masked, chrom_bins, mask, corrected, components, mean, masked_bins = wt.prepReference(
        None, counts=counts, chrom_bins=chrom_bins_in)
    print('Zero mask on the GPU: %d bins x %d samples -> %d bins kept, 3 PCA components removed'
          % (len(mask), n_given, masked.shape[0]))
    running = np.cumsum(masked_bins)
    stored = dict(arguments=vars(args),
                  runtime=getRuntime(),
                  binsize=binsize,
                  chromosomeBins=chrom_bins,
                  maskedData=masked,
                  mask=mask,
                  maskedChromBins=masked_bins,
                  maskedChromBinSums=[int(v) for v in running],
                  correctedData=corrected,
                  pca_components=components,
                  pca_mean=mean)
    write_npz(args.prepfile, temporary, **stored)
    return stored


def toolNewrefPart(args):
    """`newrefpart prep part m n`: reference bins for row part m of n (1-based) -> `<part>_m.npz`."""
    number, parts = args.part
    if number > parts:
        print('ERROR: part number %d exceeds the part count %d' % (number, parts))
        sys.exit(1)
    if number < 0:
        print('ERROR: part number %d is negative' % number)
        sys.exit(1)
    prep = _open_npz(args.prepfile)
    began = time.time()
    indexes, distances = wt.getReference(prep['correctedData'], prep['maskedChromBins'],
                                         prep['maskedChromBinSums'], selectRefAmount=args.refsize,
                                         part=number, splitParts=parts)
    print('part %d of %d: %d rows in %.2f s' % (number, parts, indexes.shape[0], time.time() - began))
    save_part(args.partfile, number, parts, indexes, distances, args)


def toolNewrefPost(args):
    """`newrefpost prep part n out`: stack the n part files in order, add the prep file's layout."""
    prep = _open_npz(args.prepfile)
    idx_blocks, dst_blocks = [], []
    for number in range(1, args.parts + 1):
        name = BuildFiles.part_name(args.partfile, number)
        piece = _open_npz(name)
        idx_blocks.append(np.asarray(piece['indexes']))
        dst_blocks.append(np.asarray(piece['distances']))
        print('merged', name, idx_blocks[-1].shape)
    width = max(b.shape[1] for b in idx_blocks if b.ndim == 2) if any(b.ndim == 2 for b in idx_blocks) else 0
    idx_blocks = [b.reshape(-1, width) for b in idx_blocks]
    dst_blocks = [b.reshape(-1, width) for b in dst_blocks]
    write_npz(args.outfile,
              arguments=vars(args),
              runtime=getRuntime(),
              binsize=prep['binsize'].item(),
              indexes=np.concatenate(idx_blocks, axis=0),
              distances=np.concatenate(dst_blocks, axis=0),
              chromosome_sizes=prep['chromosomeBins'],
              mask=prep['mask'],
              masked_sizes=prep['maskedChromBins'],
              pca_components=prep['pca_components'],
              pca_mean=prep['pca_mean'])


def toolNewref(args):
    """`newref`: prep, every missing part, merge, clean up (upstream wisecondor.py:30-69).

    The upstream tool fans the parts over `-cpus` worker processes; here the rows are computed
    in one pass by one process per GPU (`-gpus`, default min(-cpus, visible GPUs)) and cut into
    the same part files, so an interrupted run resumes from whatever files exist."""
    names = BuildFiles(args.outfile)
    args.prepfile, args.partfile = names.prep, names.part_base
    args.parts = max(args.parts, args.cpus)
    ranks = _rank_count(args)
    todo = _missing_parts(args.partfile, args.parts)
    if ranks > 1 and (todo or not os.path.isfile(args.prepfile)):
        from . import ranks as rk
        rk.launch(ranks, dict(job='newref', prepfile=args.prepfile, partfile=args.partfile,
                              parts=args.parts, refsize=args.refsize, infiles=list(args.infiles),
                              binsize=args.binsize, arguments=_plain_arguments(args)))
    else:
        prep = None
        if not os.path.isfile(args.prepfile):
            prep = toolNewrefPrep(args, temporary=True)        # (the file is for a resumed run; this one keeps the arrays)
        if todo:
            began = time.time()
            indexes, distances, _ = select_all_rows(prep if prep is not None else args.prepfile, args.refsize)
            print('reference bins for %d rows in %.2f s' % (indexes.shape[0], time.time() - began))
            for number in todo:
                lo, hi = wt.getPart(number - 1, args.parts, indexes.shape[0])
                save_part(args.partfile, number, args.parts, indexes[lo:hi], distances[lo:hi], args, temporary=True)
    toolNewrefPost(args)
    os.remove(args.prepfile)
    for number in range(1, args.parts + 1):
        os.remove(BuildFiles.part_name(args.partfile, number))


def _plain_arguments(args):
    """vars(args) without the function object (for hand-over to worker processes as JSON)."""
    return {k: v for k, v in vars(args).items() if k != 'func'}


# ------------------------------------------------------------------------ test ----
def zThreshold(masked_sizes, multitest, minzscore):
    """Per-bin |z| cut of the first testing cycles: `-minzscore` if given, else the normal
    quantile for one expected false positive in bins/2 * multitest tests (wisecondor.py:203-207)."""
    if minzscore is not None:
        return minzscore
    tests = sum(masked_sizes) * 0.5 * multitest
    return norm.ppf(1 - 1. / tests)


def writeTestOutput(outfile, args, binsize, result, z_threshold):
    """The `test` output file (keys and shapes of SURVEY.md App. B): per-chromosome z and
    ratio-1 arrays as object arrays, calls [n, 5] (shape (0,) when empty), the thresholds."""
    calls = np.asarray(result['results_calls'])
    if calls.size == 0:
        calls = np.array([])
    write_npz(outfile,
              arguments=vars(args),
              runtime=getRuntime(),
              binsize=binsize,
              results_r=_as_object_array(result['results_r']),
              results_z=_as_object_array(result['results_z']),
              results_cwz=result['results_cwz'],
              results_calls=calls,
              threshold_z=z_threshold,
              asdef=result['asdef'],
              aasdef=result['asdef'] * z_threshold)


def _reference_and_threshold(args, device=0):
    stored = _open_npz(args.reference)
    reference = wt.Reference.from_npz(stored, device=device)
    cut = zThreshold([int(v) for v in stored['masked_sizes']], args.multitest, args.minzscore)
    print('z-score threshold per bin:', cut)
    return reference, cut


def toolTest(args):
    """`test sample out reference`: one sample against a reference (wisecondor.py:174-281)."""
    reference, cut = _reference_and_threshold(args)
    from . import ingest
    sample = ingest.read_sample(args.infile, reference.binsize)[0]
    began = time.time()
    result = wt.test_batch(reference, [sample], cut, minrefbins=args.minrefbins, repeats=args.repeats,
                           chromosomes=list(args.chromosomes), mineffectsize=args.mineffectsize)[0]
    print('ASDES:', result['asdef'])
    print('AASDEF:', result['asdef'] * cut)
    print('z-scores and segments took %.3f s' % (time.time() - began))
    writeTestOutput(args.outfile, args, reference.binsize, result, cut)
    reference.close()
    sys.exit(0)


def toolTestBatch(args):
    """`testbatch samples... outdir reference` (build-only): the same numbers and the same
    per-sample files as one `test` per sample, in GPU batches with pooled file decode / encode.
    `-gpus N` (or a torchrun environment) shards the sample list over N processes."""
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if world_env == 1 and _rank_count(args) > 1:
        from . import ranks as rk
        rk.launch(_rank_count(args), dict(job='testbatch', arguments=_plain_arguments(args)))
        return
    from . import ingest
    from .distributed import shard_samples
    rank = int(os.environ.get('RANK', '0'))
    device = int(os.environ.get('LOCAL_RANK', '0')) % max(1, _visible_gpus())   # functional runs may share a GPU
    reference, cut = _reference_and_threshold(args, device=device)
    lo, hi = shard_samples(len(args.infiles), rank, world_env)
    os.makedirs(args.outdir, exist_ok=True)
    stats = ingest.run_testbatch(reference, args.infiles[lo:hi], args.outdir, cut, args, runtime=getRuntime())
    print('rank %d: %d samples in %.2f s (%.1f files/s end to end; GPU batches %.3f s = %.3f waiting for the result '
          'writers + %.3f copy in and kernels + %.3f results out)'
          % (rank, stats['files'], stats['wall_s'], stats['files_per_s'], stats['gpu_s'],
             stats.get('wait_writers_s', 0.0), stats.get('h2d_and_kernels_s', 0.0), stats.get('d2h_s', 0.0)))
    reference.close()


# ---------------------------------------------------------------------- parser ----
def _not_in_this_build(args):
    print('ERROR: this sub-command is not part of the MI355X build (newref*, test only); '
          'run it with the upstream wisecondor.py')
    sys.exit(2)


def _chromosome_list(text):
    return [int(token) for token in text.split(',')]


#: options shared by `test` and `testbatch`: (flag, argparse settings)
_TEST_OPTIONS = (
    ('-minzscore', dict(type=float, default=None,
                        help='fixed |z| cut per bin instead of the one derived from -multitest')),
    ('-chromosomes', dict(type=_chromosome_list, default=list(range(1, 23)),
                          help='autosomes to segment, comma separated (default 1..22)')),
    ('-mineffectsize', dict(type=float, default=0,
                            help='ignore windows whose median ratio deviates less than this from 1')),
    ('-multitest', dict(type=float, default=1000,
                        help='number of samples the false-positive budget is spread over')),
    ('-minrefbins', dict(type=int, default=25,
                         help='bins with fewer usable reference bins are left out')),
    ('-repeats', dict(type=int, default=5,
                      help='z-score passes, aberrant bins masked between passes')),
)


def buildParser():
    parser = argparse.ArgumentParser(
        description='WISECONDOR newref / test on AMD MI355X (drop-in for the upstream sub-commands)')
    sub = parser.add_subparsers()

    for name in ('convert', 'plot', 'report'):
        p = sub.add_parser(name, description='not provided by this build')
        p.add_argument('rest', nargs=argparse.REMAINDER)
        p.set_defaults(func=_not_in_this_build)

    p = sub.add_parser('newref', description='Build a reference from healthy samples in one go')
    p.add_argument('infiles', type=str, nargs='*', help='converted reference samples (.npz)')
    p.add_argument('outfile', type=str, help='reference file to write (.npz)')
    p.add_argument('-refsize', type=int, default=100, help='reference bins kept per target bin')
    p.add_argument('-binsize', type=int, default=None,
                   help='merge sample bins to this size first (a multiple of their own size)')
    p.add_argument('-cpus', type=int, default=1,
                   help='upstream worker count: sets the number of part files and, up to the '
                        'number of visible GPUs, of GPU processes')
    p.add_argument('-parts', type=int, default=1, help='number of part files when larger than -cpus')
    p.add_argument('-gpus', type=int, default=None, help='GPU processes to use (build-only option)')
    p.set_defaults(func=toolNewref)

    p = sub.add_parser('newrefprep', description='Step 1 of a split reference build: the prep file')
    p.add_argument('infiles', type=str, nargs='*', help='converted reference samples (.npz)')
    p.add_argument('prepfile', type=str, help='prep file to write (.npz)')
    p.add_argument('-binsize', type=int, default=None,
                   help='merge sample bins to this size first (a multiple of their own size)')
    p.set_defaults(func=toolNewrefPrep)

    p = sub.add_parser('newrefpart', description='Step 2 of a split reference build: one part')
    p.add_argument('prepfile', type=str, help='prep file of step 1')
    p.add_argument('partfile', type=str, help='base name of the part files; _<m>.npz is appended')
    p.add_argument('part', type=int, default=[0, 1], nargs=2, help='m n: compute part m of n')
    p.add_argument('-refsize', type=int, default=100, help='reference bins kept per target bin')
    p.set_defaults(func=toolNewrefPart)

    p = sub.add_parser('newrefpost', description='Step 3 of a split reference build: the merge')
    p.add_argument('prepfile', type=str, help='prep file of step 1')
    p.add_argument('partfile', type=str, help='base name of the part files of step 2')
    p.add_argument('parts', type=int, default=1, help='how many parts step 2 was split into')
    p.add_argument('outfile', type=str, help='reference file to write (.npz)')
    p.set_defaults(func=toolNewrefPost)

    p = sub.add_parser('test', description='Call copy number aberrations in one sample')
    p.add_argument('infile', type=str, help='converted sample (.npz)')
    p.add_argument('outfile', type=str, help='result file to write (.npz)')
    p.add_argument('reference', type=str, help='reference built by newref')
    for flag, settings in _TEST_OPTIONS:
        p.add_argument(flag, **settings)
    p.set_defaults(func=toolTest)

    p = sub.add_parser('testbatch', description='test for many samples per GPU batch (build-only)')
    p.add_argument('infiles', type=str, nargs='+', help='converted samples (.npz)')
    p.add_argument('outdir', type=str, help='directory receiving <sample>_test.npz')
    p.add_argument('reference', type=str, help='reference built by newref')
    p.add_argument('-batch', type=int, default=256, help='samples per GPU batch')
    p.add_argument('-gpus', type=int, default=None, help='GPU processes to shard the samples over')
    p.add_argument('-io', type=int, default=8, help='file decode / encode threads')
    p.add_argument('-ziplevel', type=int, default=1,
                   help='zlib level of the result files (0 = stored, 1 = run-length deflate: the same size at a third of '
                        'the time on these arrays, 6 = what np.savez_compressed uses)')
    for flag, settings in _TEST_OPTIONS:
        p.add_argument(flag, **settings)
    p.set_defaults(func=toolTestBatch)
    return parser


def main(argv=None):
    parser = buildParser()
    args = parser.parse_args(sys.argv[1:] if argv is None else argv)
    if not hasattr(args, 'func'):
        parser.print_usage()
        sys.exit(2)
    printArgs(args)
    args.func(args)


if __name__ == '__main__':
    main()
