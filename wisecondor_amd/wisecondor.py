"""Drop-in `wisecondor.py` command line for the newref* and test sub-commands.

Same sub-commands, positional arguments, single-dash options, defaults, file
naming and .npz keys as the reference CLI (wisecondor.py:345-521); the numeric
work runs on an MI355X through libwisecondor_hip.so.  convert / plot / report
are outside this build's scope and exit with a message.
"""
import argparse
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

curTime = datetime.datetime.now()


def getVersion():
    version = 'unknown'
    try:
        version = subprocess.check_output(["git", "describe", "--always"], stderr=subprocess.DEVNULL).split()[0]
    except Exception:
        pass
    return version


def getRuntime():
    """Provenance dict stored in every output (wisetools.py:47-53)."""
    return dict(version=getVersion(), datetime=curTime, hostname=socket.gethostname(),
                username=getpass.getuser())


def printArgs(args):
    argdict = vars(args)
    print('tool =', str(argdict['func']).split()[1][4:])
    for arg in sorted(argdict.keys()):
        if arg != 'func':
            print(arg, '=', argdict[arg])


def _load(path):
    return np.load(path, allow_pickle=True, encoding='latin1')


def _object_array(items):
    out = np.empty(len(items), dtype=object)
    for i, v in enumerate(items):
        out[i] = v
    return out


def toolNewref(args):
    """prep -> parts -> post with resume-by-file-existence (wisecondor.py:30-69)."""
    splitPath = list(os.path.split(args.outfile))
    if splitPath[-1][-4:] == '.npz':
        splitPath[-1] = splitPath[-1][:-4]
    basePath = os.path.join(splitPath[0], splitPath[1])
    args.prepfile = basePath + "_prep.npz"
    args.partfile = basePath + "_part"
    args.parts = max(args.parts, args.cpus)

    if not os.path.isfile(args.prepfile):
        toolNewrefPrep(args)
    # -cpus asked the reference for a process pool; one GPU runs the parts back to back
    for part in range(1, args.parts + 1):
        if not os.path.isfile(args.partfile + "_" + str(part) + ".npz"):
            args.part = [part, args.parts]
            toolNewrefPart(args)
    toolNewrefPost(args)
    os.remove(args.prepfile)
    for part in range(1, args.parts + 1):
        os.remove(args.partfile + '_' + str(part) + '.npz')


def toolNewrefPrep(args):
    """wisecondor.py:72-108; normalisation, mask and PCA run through wt.prepReference (GPU)."""
    samples = []
    binsizes = set()
    for infile in args.infiles:
        print('Loading:', infile, end=' ')
        npzdata = _load(infile)
        binsize = npzdata['arguments'].item()['binsize']
        print(' \tbinsize:', int(binsize))
        samples.append(wt.scaleSample(npzdata['sample'].item(), binsize, args.binsize))
        binsizes.add(binsize)

    if args.binsize is None and len(binsizes) != 1:
        print('ERROR: There appears to be a mismatch in binsizes in your dataset:', binsizes)
        print('Either remove the offending sample or use -binsize to scale all samples')
        sys.exit(1)

    binsize = args.binsize
    if args.binsize is None:
        binsize = binsizes.pop()

    print('Applying nonzero mask on the data and fitting the PCA on the GPU:', end=' ')
    maskedData, chromosomeBins, mask, correctedData, comps, mean, maskedChromBins = wt.prepReference(samples)
    print((len(mask), len(samples)), 'becomes', maskedData.shape)
    del samples
    maskedChromBinSums = [int(v) for v in np.cumsum(maskedChromBins)]
    pca = wt._PCAResult(comps, mean)
    np.savez_compressed(args.prepfile,
                        arguments=vars(args),
                        runtime=getRuntime(),
                        binsize=binsize,
                        chromosomeBins=chromosomeBins,
                        maskedData=maskedData,
                        mask=mask,
                        maskedChromBins=maskedChromBins,
                        maskedChromBinSums=maskedChromBinSums,
                        correctedData=correctedData,
                        pca_components=pca.components_,
                        pca_mean=pca.mean_)


def toolNewrefPart(args):
    """wisecondor.py:111-132: reference-bin selection for one part, on the GPU."""
    if args.part[0] > args.part[1]:
        print('ERROR: Part should be smaller or equal to total parts:', args.part[0], '>', args.part[1], 'is wrong')
        sys.exit(1)
    if args.part[0] < 0:
        print('ERROR: Part should be at least zero:', args.part[0], '<', 0, 'is wrong')
        sys.exit(1)

    npzdata = _load(args.prepfile)
    correctedData = npzdata['correctedData']
    maskedChromBins = npzdata['maskedChromBins']
    maskedChromBinSums = npzdata['maskedChromBinSums']

    start = time.time()
    indexes, distances = wt.getReference(correctedData, maskedChromBins, maskedChromBinSums,
                                         selectRefAmount=args.refsize, part=args.part[0],
                                         splitParts=args.part[1])
    print(args.part[0], 'Time spent:', int(time.time() - start), 'seconds')

    np.savez_compressed(args.partfile + '_' + str(args.part[0]) + '.npz',
                        arguments=vars(args),
                        runtime=getRuntime(),
                        indexes=indexes,
                        distances=distances)


def toolNewrefPost(args):
    """wisecondor.py:135-170."""
    npzdata = _load(args.prepfile)
    maskedChromBins = npzdata['maskedChromBins']
    chromosomeBins = npzdata['chromosomeBins']
    mask = npzdata['mask']
    pca_components = npzdata['pca_components']
    pca_mean = npzdata['pca_mean']
    binsize = npzdata['binsize'].item()

    bigIndexes = []
    bigDistances = []
    for part in range(1, args.parts + 1):
        infile = args.partfile + '_' + str(part) + '.npz'
        print('Loading:', infile)
        npzdata = _load(infile)
        bigIndexes.extend(npzdata['indexes'])
        bigDistances.extend(npzdata['distances'])
        print(part, npzdata['indexes'].shape)

    indexes = np.array(bigIndexes)
    distances = np.array(bigDistances)

    np.savez_compressed(args.outfile,
                        arguments=vars(args),
                        runtime=getRuntime(),
                        binsize=binsize,
                        indexes=indexes,
                        distances=distances,
                        chromosome_sizes=chromosomeBins,
                        mask=mask,
                        masked_sizes=maskedChromBins,
                        pca_components=pca_components,
                        pca_mean=pca_mean)


def zThreshold(masked_sizes, multitest, minzscore):
    """wisecondor.py:203-207."""
    num_tests = sum(masked_sizes)
    z_threshold = norm.ppf(1 - 1. / (num_tests * 0.5 * multitest))
    if minzscore is not None:
        z_threshold = minzscore
    return z_threshold


def writeTestOutput(outfile, args, binsize, out, z_threshold):
    """The test .npz exactly as the reference lays it out (wisecondor.py:270-280): ragged
    per-chromosome lists become object arrays (modern numpy refuses to infer them), an empty
    call list stays shape (0,) like np.array([])."""
    calls = np.asarray(out['results_calls'])
    stdDevAvg = out['asdef']
    np.savez_compressed(outfile,
                        arguments=vars(args),
                        runtime=getRuntime(),
                        binsize=binsize,
                        results_r=_object_array(out['results_r']),
                        results_z=_object_array(out['results_z']),
                        results_cwz=out['results_cwz'],
                        results_calls=calls if len(calls) else np.array([]),
                        threshold_z=z_threshold,
                        asdef=stdDevAvg,
                        aasdef=stdDevAvg * z_threshold)


def toolTest(args):
    """wisecondor.py:174-281: one sample against a reference, on the GPU."""
    referenceFile = _load(args.reference)
    reference = wt.Reference.from_npz(referenceFile)
    binsize = reference.binsize
    masked_sizes = [int(v) for v in referenceFile['masked_sizes']]
    del referenceFile

    sampleFile = _load(args.infile)
    sample = sampleFile['sample'].item()
    sampleBinSize = sampleFile['arguments'].item()['binsize']
    sample = wt.scaleSample(sample, sampleBinSize, binsize)

    z_threshold = zThreshold(masked_sizes, args.multitest, args.minzscore)
    print('Per bin z-score threshold for first testing cycles:', z_threshold)

    start = time.time()
    out = wt.test_batch(reference, [sample], z_threshold, minrefbins=args.minrefbins,
                        repeats=args.repeats, chromosomes=list(args.chromosomes),
                        mineffectsize=args.mineffectsize)[0]
    stdDevAvg = out['asdef']
    print('ASDES:', stdDevAvg, '\nAASDEF:', stdDevAvg * z_threshold)
    print('Time spent on z-scores and stouffers z-scores:', int(time.time() - start), 'seconds')

    writeTestOutput(args.outfile, args, binsize, out, z_threshold)
    reference.close()
    sys.exit(0)


def toolTestBatch(args):
    """Build-only addition: `test` for many samples in one GPU batch (same numbers and the
    same per-sample output files as running `test` once per sample).  Under torchrun the
    samples are sharded over the ranks (one process per GPU, no collective needed)."""
    from .distributed import shard_samples
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    device = int(os.environ.get('LOCAL_RANK', '0'))
    referenceFile = _load(args.reference)
    reference = wt.Reference.from_npz(referenceFile, device=device)
    binsize = reference.binsize
    masked_sizes = [int(v) for v in referenceFile['masked_sizes']]
    del referenceFile
    z_threshold = zThreshold(masked_sizes, args.multitest, args.minzscore)
    print('Per bin z-score threshold for first testing cycles:', z_threshold)
    lo, hi = shard_samples(len(args.infiles), rank, world)
    infiles = args.infiles[lo:hi]
    if not os.path.isdir(args.outdir):
        os.makedirs(args.outdir, exist_ok=True)
    start = time.time()
    for at in range(0, len(infiles), args.batch):
        names = infiles[at:at + args.batch]
        samples = []
        for name in names:
            sampleFile = _load(name)
            samples.append(wt.scaleSample(sampleFile['sample'].item(),
                                          sampleFile['arguments'].item()['binsize'], binsize))
        outs = wt.test_batch(reference, samples, z_threshold, minrefbins=args.minrefbins, repeats=args.repeats,
                             chromosomes=list(args.chromosomes), mineffectsize=args.mineffectsize)
        for name, out in zip(names, outs):
            base = os.path.basename(name)
            base = base[:-4] if base.endswith('.npz') else base
            one = argparse.Namespace(**vars(args))
            one.infile = name
            one.outfile = os.path.join(args.outdir, base + '_test.npz')
            writeTestOutput(one.outfile, one, binsize, out, z_threshold)
            print(name, '->', one.outfile, 'calls:', len(out['results_calls']))
    print('Time spent on', len(infiles), 'samples:', int(time.time() - start), 'seconds')
    reference.close()


def _out_of_scope(args):
    print('ERROR: this sub-command is not part of the MI355X build (newref*, test only); '
          'use the upstream wisecondor.py for it')
    sys.exit(2)


def _int_list(x):
    return [int(v) for v in x.split(',')]


def buildParser():
    parser = argparse.ArgumentParser(
        description="WISECONDOR (WIthin-SamplE COpy Number aberration DetectOR) -- MI355X build")
    subparsers = parser.add_subparsers()

    for name in ('convert', 'plot', 'report'):
        p = subparsers.add_parser(name, description='not provided by this build')
        p.add_argument('rest', nargs=argparse.REMAINDER)
        p.set_defaults(func=_out_of_scope)

    parser_newref = subparsers.add_parser('newref',
        description='Create a new reference using healthy reference samples')
    parser_newref.add_argument('infiles', type=str, nargs='*',
        help='Path and all to reference data files (i.e. ./reference/*.npz)')
    parser_newref.add_argument('outfile', type=str,
        help='Path and filename for the reference output (i.e. ./reference/myref.npz)')
    parser_newref.add_argument('-refsize', type=int, default=100,
        help='Amount of reference locations per target')
    parser_newref.add_argument('-binsize', type=int, default=None,
        help='Try to scale samples to this binsize, multiples of existing binsize only')
    parser_newref.add_argument('-cpus', type=int, default=1,
        help='Accepted for compatibility: sets the number of parts (one GPU runs them in turn)')
    parser_newref.add_argument('-parts', type=int, default=1,
        help='Split reference finding in this many jobs, only used if > cpus')
    parser_newref.set_defaults(func=toolNewref)

    parser_newrefprep = subparsers.add_parser('newrefprep',
        description='Prepare creation of new reference split over several processes')
    parser_newrefprep.add_argument('infiles', type=str, nargs='*',
        help='Path and all to reference data files (i.e. ./reference/*.npz)')
    parser_newrefprep.add_argument('prepfile', type=str,
        help='Path and filename for the prep output (i.e. ./reference/myref_prep.npz)')
    parser_newrefprep.add_argument('-binsize', type=int, default=None,
        help='Try to scale samples to this binsize, multiples of existing binsize only')
    parser_newrefprep.set_defaults(func=toolNewrefPrep)

    parser_newrefpart = subparsers.add_parser('newrefpart',
        description='Creation of new reference split over several processes')
    parser_newrefpart.add_argument('prepfile', type=str,
        help='Path and filename for the prep file  (i.e. ./reference/myref_prep.npz)')
    parser_newrefpart.add_argument('partfile', type=str,
        help='Path and basename for the reference part output, receives _m.npz extension from part argument')
    parser_newrefpart.add_argument('part', type=int, default=[0, 1], nargs=2,
        help='Part m out of n parts, appends _m.npz to file name')
    parser_newrefpart.add_argument('-refsize', type=int, default=100,
        help='Amount of reference locations per target')
    parser_newrefpart.set_defaults(func=toolNewrefPart)

    parser_newrefpost = subparsers.add_parser('newrefpost',
        description='Combine creation of new reference split over several processes')
    parser_newrefpost.add_argument('prepfile', type=str,
        help='Path and filename for the prep file  (i.e. ./reference/myref_prep.npz)')
    parser_newrefpost.add_argument('partfile', type=str,
        help='Path and basename to reference part data files, appends _m.npz depending on parts')
    parser_newrefpost.add_argument('parts', type=int, default=1,
        help='Used combine all data parts, represents n parts previously specified')
    parser_newrefpost.add_argument('outfile', type=str,
        help='Path and filename for the reference output (i.e. ./reference/myref.npz)')
    parser_newrefpost.set_defaults(func=toolNewrefPost)

    parser_test = subparsers.add_parser('test', description='Test sample for Copy Number Aberrations')
    parser_test.add_argument('infile', type=str, help='Sample to test')
    parser_test.add_argument('outfile', type=str, help='Basename of files to write')
    parser_test.add_argument('reference', type=str, help='Reference as previously created')
    parser_test.add_argument('-minzscore', type=float, default=None, help='Minimum absolute z-score')
    parser_test.add_argument('-chromosomes', help="Integer of every chromosome to test, comma delimited",
                             type=_int_list, default=list(range(1, 23)))
    parser_test.add_argument('-mineffectsize', type=float, default=0,
        help='Minimum absolute relative change in read depth')
    parser_test.add_argument('-multitest', type=float, default=1000,
        help='Increase z-score to compensate for multiple sample testing')
    parser_test.add_argument('-minrefbins', type=int, default=25,
        help='Minimum amount of sensible ref bins per target bin')
    parser_test.add_argument('-repeats', type=int, default=5, help='Repeats when calling')
    parser_test.set_defaults(func=toolTest)

    # build-only addition: the same test for many samples per GPU batch
    parser_tb = subparsers.add_parser('testbatch', description='Test many samples in GPU batches')
    parser_tb.add_argument('infiles', type=str, nargs='+', help='Samples to test')
    parser_tb.add_argument('outdir', type=str, help='Directory for the <sample>_test.npz outputs')
    parser_tb.add_argument('reference', type=str, help='Reference as previously created')
    parser_tb.add_argument('-batch', type=int, default=256, help='Samples per GPU batch')
    parser_tb.add_argument('-minzscore', type=float, default=None, help='Minimum absolute z-score')
    parser_tb.add_argument('-chromosomes', type=_int_list, default=list(range(1, 23)),
                           help='Chromosomes to test, comma delimited')
    parser_tb.add_argument('-mineffectsize', type=float, default=0,
                           help='Minimum absolute relative change in read depth')
    parser_tb.add_argument('-multitest', type=float, default=1000,
                           help='Increase z-score to compensate for multiple sample testing')
    parser_tb.add_argument('-minrefbins', type=int, default=25,
                           help='Minimum amount of sensible ref bins per target bin')
    parser_tb.add_argument('-repeats', type=int, default=5, help='Repeats when calling')
    parser_tb.set_defaults(func=toolTestBatch)
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
