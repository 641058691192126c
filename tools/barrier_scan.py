#!/usr/bin/env python3
"""Every s_barrier of a gfx950 assembly listing must be reached with this wave's LDS operations complete: walking back
from the barrier, an `s_waitcnt ... lgkmcnt(0)` has to come before any DS / FLAT / scalar-memory instruction AND before
the head of the basic block (a label: other paths join there and their pending operations are not visible in this
listing).  Round 6 found `__syncthreads()` at the head of k_seg_walk's loop compiled to a bare s_barrier behind back
edges that carry ds_write instructions (the compiler's wait insertion lost the release's lgkmcnt(0) there).

    python tools/barrier_scan.py file.s [file.s ...]     -> prints kernel, line and reason of every unguarded barrier
"""
import re
import sys


def scan(path):
    kernel = "?"
    block = []          # instructions of the current basic block
    bad = []
    total = 0
    for no, raw in enumerate(open(path, errors="ignore"), 1):
        t = raw.strip()
        if not t or t.startswith(";"):
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", t)
        if m:
            if not m.group(1).startswith(".L"):
                kernel = m.group(1)
            block = []
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        if op == "s_barrier":
            total += 1
            reason = "block head reached without a wait (paths join here)"
            for prev in reversed(block):
                pop = prev.split()[0]
                if pop == "s_waitcnt" and ("lgkmcnt(0)" in prev):
                    reason = None
                    break
                if pop == "s_waitcnt" and re.fullmatch(r"s_waitcnt\s+0(x0+)?", prev):
                    reason = None
                    break
                if pop.startswith("ds_bpermute") or pop.startswith("ds_permute") or pop.startswith("ds_swizzle"):
                    continue            # lane exchanges through the LDS crossbar: no memory is written
                if pop.startswith("ds_") or pop.startswith("flat_") or pop.startswith("s_load") or pop.startswith("s_buffer_load") \
                        or pop.startswith("s_store") or pop.startswith("s_atomic"):
                    reason = "%s after the last wait" % pop
                    break
            if reason:
                bad.append((kernel, no, reason))
        block.append(t)
    return total, bad


if __name__ == "__main__":
    rc = 0
    for path in sys.argv[1:]:
        total, bad = scan(path)
        print("%s: %d barriers, %d unguarded" % (path, total, len(bad)))
        for kernel, no, reason in bad:
            print("   %s line %d: %s" % (kernel[:70], no, reason))
            rc = 1
    sys.exit(rc)
