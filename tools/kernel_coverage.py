#!/usr/bin/env python3
"""Which kernels of libwisecondor_hip.so did a profiled run launch?

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/cov -o c -- python3 -m pytest $REPO/tests -m gpu -q -p no:cacheprovider
    python3 tools/kernel_coverage.py gpurun_out/cov

Lists every kernel symbol of the library (amdhsa kernel descriptors, demangled) and marks the ones that appear in
the run's kernel statistics; exit status 1 if one never ran."""
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def library_kernels():
    """Kernel names from the device assembly of every .hip source (the same flags as the build)."""
    sys.path.insert(0, ROOT)
    from wisecondor_amd.build import FLAGS, SOURCES, CSRC, _hipcc
    names = set()
    for src in SOURCES:
        if not src.endswith(".hip"):
            continue
        out = "/tmp/wc_cov_%s.s" % src
        flags = [f for f in FLAGS if f != "-fPIC"]
        subprocess.check_call([_hipcc()] + flags + ["--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, src)],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
        for m in re.finditer(r"^\s+\.amdhsa_kernel\s+(\S+)", text, re.M):
            names.add(m.group(1))
    dem = subprocess.run(["c++filt"], input="\n".join(sorted(names)), capture_output=True, text=True).stdout.splitlines()
    return sorted(set(d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for d in dem))


def launched(folder):
    seen = set()
    for root, _, files in os.walk(folder):
        for f in files:
            if f.endswith("kernel_stats.csv"):
                for row in csv.DictReader(open(os.path.join(root, f))):
                    seen.add(row["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0])
    return seen


def main():
    kernels, seen = library_kernels(), launched(sys.argv[1])
    missing = [k for k in kernels if k not in seen]
    for k in kernels:
        print("%-8s %s" % ("ran" if k in seen else "NEVER", k))
    print("%d kernels in the library, %d launched, %d never" % (len(kernels), len(kernels) - len(missing), len(missing)))
    return 1 if missing else 0


if __name__ == "__main__":
    sys.exit(main())
