#!/usr/bin/env python3
"""Build an experimental copy of the library from a patched newref.hip / testpath.hip:
    python tools/build_variant.py NAME FILE.hip 'old text' 'new text' ['old' 'new' ...]
-> wisecondor_amd/ab/lib_NAME.so (git-ignored; travels to the GPU box).  Select it with
WC_LIB_PATH.  The other objects are the ones of the regular build."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wisecondor_amd.build import CSRC, FLAGS, SOURCES, _hipcc  # noqa: E402

name, fname = sys.argv[1], sys.argv[2]
pairs = sys.argv[3:]
text = open(os.path.join(CSRC, fname)).read()
for old, new in zip(pairs[0::2], pairs[1::2]):
    assert text.count(old) >= 1, "not found: %r" % old
    text = text.replace(old, new)
tmp = os.path.join(CSRC, "_ab_%s_%s" % (name, fname))
open(tmp, "w").write(text)
obj = os.path.join(ROOT, "wisecondor_amd", "ab", "%s_%s.o" % (name, fname[:-4]))
try:
    subprocess.check_call([_hipcc()] + FLAGS + ["-c", tmp, "-o", obj])
finally:
    os.remove(tmp)
objs = [obj if s == fname else os.path.join(CSRC, os.path.splitext(s)[0] + ".o") for s in SOURCES]
out = os.path.join(ROOT, "wisecondor_amd", "ab", "lib_%s.so" % name)
subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lz"])
print(out)
