"""The compiled kernels themselves (gfx950 assembly, cross-compiled here): every workgroup barrier must be reached with
the wave's own LDS operations complete.

Round 6's root cause of round 5's run-to-run wrong medians / lost segments: `__syncthreads()` at the head of
k_seg_walk's loop was emitted as a bare `s_barrier` behind back edges that carry thread 0's `ds_write` of the stack
pointer (no `s_waitcnt lgkmcnt(0)`, at -O1 and -O3 alike), so other waves could read the old pointer after the barrier
and fall out of step.  Every barrier in csrc/ now goes through wc_sync() (explicit wait + __syncthreads());
tools/barrier_scan.py proves it on the listing, this test keeps it proven for the next edit.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_barrier_without_its_wait(tmp_path):
    import barrier_scan
    from wisecondor_amd.build import CSRC, FLAGS, _hipcc
    sources = ["testpath.hip", "newref.hip", "prep.hip", "eigh.hip"]
    flags = [f for f in FLAGS if f != "-fPIC"]
    procs = []
    for src in sources:
        out = str(tmp_path / (src[:-4] + ".s"))
        procs.append((src, out, subprocess.Popen([_hipcc()] + flags + ["--cuda-device-only", "-S", "-o", out,
                                                                        os.path.join(CSRC, src)],
                                                 stderr=subprocess.DEVNULL)))
    barriers = 0
    for src, out, p in procs:
        assert p.wait() == 0, src
        total, bad = barrier_scan.scan(out)
        assert total > 0, src
        assert not bad, (src, bad[:5])
        barriers += total
    assert barriers > 400


def test_sources_use_the_guarded_barrier():
    """No bare __syncthreads() call in csrc/: wc_sync() / wc_sync_or() (common.h) only."""
    import re
    csrc = os.path.join(ROOT, "wisecondor_amd", "csrc")
    for name in os.listdir(csrc):
        if not name.endswith((".hip", ".h", ".cpp")):
            continue
        text = open(os.path.join(csrc, name)).read()
        code = re.sub(r"//[^\n]*", "", text)
        calls = re.findall(r"__syncthreads(?:_or|_and|_count)?\s*\(", code)
        if name == "common.h":
            assert len(calls) == 2, calls            # inside wc_sync / wc_sync_or
        else:
            assert not calls, (name, calls)
