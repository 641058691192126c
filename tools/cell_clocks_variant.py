#!/usr/bin/env python3
"""Development aid: the library with k_seg_job's phase clocks compiled in -> wisecondor_amd/ab/lib_cellclk.so
(read with tools/gpu_cell_clocks.py on the GPU box under WC_LIB_PATH)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "cellclk", "testpath.hip",
                       "// WC_CELL_CLOCKS_SWITCH\n", "#define WC_CELL_CLOCKS 1\n"])
