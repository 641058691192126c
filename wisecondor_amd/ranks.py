"""One process per GPU for the command line tools (torch.distributed over RCCL / xGMI).

The upstream `newref -cpus N` starts N worker processes that each compute some row parts
and exchange nothing but files (wisecondor.py:47-56).  Here `launch()` starts N fresh
Python processes -- before the calling process has touched the GPU -- each bound to one GPU:

newref     every rank loads the prep file (rank 0 writes it first if it is missing), the
           ranks share one NewrefJob pass (symmetric tile shard + candidate exchange, or row
           bands; wisecondor_amd.distributed) and every rank writes the part files whose rows it
           owns -- no result all-gather -- so the merge step of the parent finds exactly the
           files the upstream tool would have left (parts that straddle two ranks' row ranges:
           every rank gathers all rows, part m goes to rank m mod N).
testbatch  the sample list is cut into N contiguous shards, no collective.

A worker that fails makes launch() raise: the upstream pool never collects its futures, so
a crashed worker only shows when the merge cannot open a part file.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(n_ranks, spec, backend=None, extra_env=None):
    """Run `spec` (a JSON-able dict with a 'job' key) on n_ranks processes; wait for all."""
    port = _free_port()
    with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as f:
        json.dump(spec, f)
        spec_path = f.name
    procs = []
    # the children import this package the way this process did, whatever the working directory is
    pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        for rank in range(n_ranks):
            env = dict(os.environ)
            env['PYTHONPATH'] = os.pathsep.join([pkg_parent] + [p for p in env.get('PYTHONPATH', '').split(os.pathsep) if p])
            env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(n_ranks),
                        'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port),
                        'HSA_ENABLE_IPC_MODE_LEGACY': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')})
            if backend:
                env['WC_RANKS_BACKEND'] = backend
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen([sys.executable, '-m', 'wisecondor_amd.ranks', spec_path], env=env))
        # a rank that dies leaves the others waiting in a collective: stop them instead of hanging
        import time
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            if any(c not in (None, 0) for c in codes):
                time.sleep(1.0)
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                codes = [p.wait() for p in procs]
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        os.unlink(spec_path)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise RuntimeError('GPU worker(s) failed: ' + ', '.join('rank %d exit %d' % rc for rc in bad))


def _init(backend):
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    n_dev = max(1, torch.cuda.device_count())
    device = int(os.environ.get('LOCAL_RANK', '0')) % n_dev      # gloo functional runs may share a GPU
    torch.cuda.set_device(device)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', device))
    else:
        dist.init_process_group(backend)
    return dist, rank, world, device


def _newref(spec, dist, rank, world, device):
    from . import wisecondor as cli
    from . import wisetools as wt
    args = argparse.Namespace(**spec['arguments'])
    args.func = cli.toolNewref
    if not os.path.isfile(spec['prepfile']):
        if rank == 0:
            cli.toolNewrefPrep(args)
    dist.barrier()
    todo = [m for m in range(1, spec['parts'] + 1)
            if not os.path.isfile(cli.BuildFiles.part_name(spec['partfile'], m))]
    dist.barrier()                       # every rank has listed the missing parts before any is written
    if not todo:
        return
    # Part m goes to the rank that owns its rows (with `parts` a multiple of the rank count every part lies inside one
    # rank's row range): the ranks then keep their own rows and the result all-gather is skipped -- the reference's
    # workers exchange nothing but files either (wisecondor.py:47-56).  Parts that straddle two ranks' ranges: every
    # rank gathers all rows and part m goes to rank m mod N.
    import numpy as np
    from .distributed import row_range
    n_bins = int(np.sum(np.asarray(np.load(spec['prepfile'], allow_pickle=True)['maskedChromBins'])))
    owners = cli.part_owners(spec['parts'], todo, n_bins, world)
    indexes, distances, job = cli.select_all_rows(spec['prepfile'], spec['refsize'], device=device,
                                                  rank=rank, world=world, gather=owners is None)
    if rank == 0:
        print('reference bins for %d rows on %d GPUs (%s shard, %s)'
              % (n_bins, world, job.mode, 'parts written by the owners of their rows' if owners is not None
                 else 'every rank gathers all rows'))
    first = 0 if owners is None else row_range(rank, world, n_bins)[0]
    for m in todo:
        if (m % world if owners is None else owners[m]) != rank:
            continue
        lo, hi = wt.getPart(m - 1, spec['parts'], n_bins)
        cli.save_part(spec['partfile'], m, spec['parts'], indexes[lo - first:hi - first], distances[lo - first:hi - first], args)
    dist.barrier()


def _testbatch(spec, dist, rank, world, device):
    from . import wisecondor as cli
    args = argparse.Namespace(**spec['arguments'])
    args.func = cli.toolTestBatch
    args.gpus = 1
    cli.toolTestBatch(args)              # reads RANK / WORLD_SIZE / LOCAL_RANK for its shard
    dist.barrier()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    with open(argv[0]) as f:
        spec = json.load(f)
    backend = os.environ.get('WC_RANKS_BACKEND', 'nccl')
    dist, rank, world, device = _init(backend)
    try:
        {'newref': _newref, 'testbatch': _testbatch}[spec['job']](spec, dist, rank, world, device)
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
