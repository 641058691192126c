#!/usr/bin/env python3
"""Development aid: the library with phase clocks in k_seg_walk (thread 0 of every workgroup books clock64 deltas per
phase in LDS -- the CJ_CLK slots of cell_search plus the walker's own -- and adds them to g_dbg when it leaves)
-> wisecondor_amd/ab/lib_walkclk.so; read with tools/gpu_walk_clocks.py on the GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [
    ("// WC_CELL_CLOCKS_SWITCH\n", "#define WC_CELL_CLOCKS 1\n"),
    # walker: clocks start
    ("    const int lane = tid & 63, w = tid >> 6;\n    int wins = 0, evals = 0;\n    if (tid == 0) {\n        Job root;\n        root.region = region; root.lo = 0; root.hi = rg.n; root.pad = 0;",
     "    const int lane = tid & 63, w = tid >> 6;\n    int wins = 0, evals = 0;\n    if (tid < 32) sh.clk[tid] = 0ull;\n    if (tid == 0) sh.t_prev = clock64();\n    if (tid == 0) {\n        Job root;\n        root.region = region; root.lo = 0; root.hi = rg.n; root.pad = 0;"),
    ("        cell_seed(sh, g, T, tid);\n        wc_sync();\n        double vmax = -INFINITY, vmin = INFINITY, d2 = -INFINITY, d3 = INFINITY;\n        cell_search<0>(sh, g, rs, eps2, INFINITY, -INFINITY, nullptr, 0, 1, vmax, vmin, wins, evals, tid);\n        block_minmax4(vmax, vmin, d2, d3, tid);\n        if (sh.lost) {",
     "        cell_seed(sh, g, T, tid);\n        wc_sync();\n        CJ_CLK(1);\n        if (tid == 0) sh.clk[17] += 1ull;\n        double vmax = -INFINITY, vmin = INFINITY, d2 = -INFINITY, d3 = INFINITY;\n        cell_search<0>(sh, g, rs, eps2, INFINITY, -INFINITY, nullptr, 0, 1, vmax, vmin, wins, evals, tid);\n        block_minmax4(vmax, vmin, d2, d3, tid);\n        CJ_CLK(7);\n        if (sh.lost) {"),
    ("            if (lane == 0) s_best[w] = b;\n        }\n        wc_sync();\n        if (tid == 0) {\n            BestPair b = s_best[0];",
     "            if (lane == 0) s_best[w] = b;\n        }\n        wc_sync();\n        CJ_CLK(18);\n        if (tid == 0) {\n            BestPair b = s_best[0];"),
    ("    // ---- the region's segments, appended to the batch's list for k_walk_rows",
     "    wc_sync();\n    CJ_CLK(19);\n    if (tid == 0) sh.clk[16] = 1ull;\n    wc_sync();\n    if (tid < 32) atomicAdd(&g_dbg[tid], sh.clk[tid]);\n    // ---- the region's segments, appended to the batch's list for k_walk_rows"),
]
args = [sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "walkclk", "testpath.hip"]
for a, b in PAIRS:
    args += [a, b]
subprocess.check_call(args)
