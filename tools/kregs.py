#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of a .hip source (device-only assembly, gfx950):
   python tools/kregs.py wisecondor_amd/csrc/newref.hip [name filter]"""
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, ".")
from wisecondor_amd.build import FLAGS, _hipcc  # noqa: E402

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = tempfile.mktemp(suffix=".s")
flags = [f for f in FLAGS if f != "-fPIC"]
subprocess.check_call([_hipcc()] + flags + ["--cuda-device-only", "-S", "-o", out, src], stderr=subprocess.DEVNULL)
text = open(out).read()
meta = text[text.index("amdhsa.kernels:"):]
for block in meta.split("  - .agpr_count:")[1:]:
    block = ".agpr_count:" + block
    get = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, block) or [None, "?"])[1]
    name = get("name")
    name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    if flt in name:
        print("%-48s vgpr %4s agpr %4s spill %3s scratch %5s lds %6s sgpr %3s" % (
            name[:48], get("vgpr_count"), get("agpr_count"), get("vgpr_spill_count"),
            get("private_segment_fixed_size"), get("group_segment_fixed_size"), get("sgpr_count")))
