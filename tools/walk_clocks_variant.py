#!/usr/bin/env python3
"""Development aid: the library with phase clocks in k_seg_walk (thread 0 of every workgroup books clock64 deltas per
phase in LDS -- the CJ_CLK slots of cell_search plus the walker's own -- and adds them to g_dbg when it leaves)
-> wisecondor_amd/ab/lib_walkclk.so; read with tools/gpu_walk_clocks.py on the GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [
    ("// WC_CELL_CLOCKS_SWITCH\n", "#define WC_CELL_CLOCKS 1\n"),
    # per-workgroup start / end (100 MHz wall clock) of the first 4096 workgroups: g_dbg[64 + 2 b], [65 + 2 b]; read with
    # wc_debug_times(ctx, -7, buffer of 64 + 8192 words)
    ("__device__ unsigned long long g_dbg[64];", "__device__ unsigned long long g_dbg[64 + 8192 + 16384];"),
    ("    if (out64) WC_HIP(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 64));",
     "    if (out64) WC_HIP(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * (block_plus_one == -7 ? 64 + 8192 + 16384 : 64)));\n    if (block_plus_one == -7) block_plus_one = 0;"),
    # walker: clocks start
    ("    int wins = 0, evals = 0;\n    if (tid0 == 0) {\n        Job root;",
     "    int wins = 0, evals = 0;\n    if (tid0 < 32) sh.clk[tid0] = 0ull;\n    if (tid0 == 0) { sh.t_prev = clock64(); if (blockIdx.x < 4096) g_dbg[64 + 2 * blockIdx.x] = wall_clock64(); }\n    if (tid0 == 0) {\n        Job root;"),
    ("        cell_seed(sh, g, T, tid);\n        wc_sync();\n        double vmax = -INFINITY, vmin = INFINITY, d2 = -INFINITY, d3 = INFINITY;\n        cell_search<0>(sh, g, rs, eps2, INFINITY, -INFINITY, nullptr, 0, 1, vmax, vmin, wins, evals, tid, true);\n        block_minmax4(vmax, vmin, d2, d3, tid);\n",
     "        cell_seed(sh, g, T, tid);\n        wc_sync();\n        CJ_CLK(1);\n        if (tid == 0) sh.clk[17] += 1ull;\n        double vmax = -INFINITY, vmin = INFINITY, d2 = -INFINITY, d3 = INFINITY;\n        cell_search<0>(sh, g, rs, eps2, INFINITY, -INFINITY, nullptr, 0, 1, vmax, vmin, wins, evals, tid, true);\n        block_minmax4(vmax, vmin, d2, d3, tid);\n        CJ_CLK(7);\n"),
    ("            if (lane == 0) s_best[w] = b;\n        }\n        wc_sync();\n        if (tid == 0) {\n            BestPair b = s_best[0];",
     "            if (lane == 0) s_best[w] = b;\n        }\n        wc_sync();\n        CJ_CLK(18);\n        if (tid == 0) {\n            BestPair b = s_best[0];"),
    ("    // ---- the region's segments, appended to the batch's list for k_walk_rows",
     "    {\n    const int tid = tid0;\n    wc_sync();\n    CJ_CLK(19);\n    if (tid == 0) sh.clk[16] = 1ull;\n    wc_sync();\n    if (tid < 32) atomicAdd(&g_dbg[tid], sh.clk[tid]);\n    if (tid == 0 && blockIdx.x < 4096) { g_dbg[65 + 2 * blockIdx.x] = wall_clock64(); unsigned long long *x_ = g_dbg + 64 + 8192 + blockIdx.x; x_[0] = sh.clk[17]; x_[4096] = (sh.clk[3] << 32) | (sh.clk[4] & 0xFFFFFFFFull); x_[8192] = sh.clk[18]; x_[12288] = (sh.clk[5] << 32) | sh.clk[6]; }\n    }\n    // ---- the region's segments, appended to the batch's list for k_walk_rows"),
    # near sweep (the product's CJ_CLK slots 20 / 22 / 23: wait + stage, bounds + pushes, queue drain); [24] trips, [25] queued pairs
    ("        const int n_items = sh.n_items < CJ_ITEMQ ? sh.n_items : CJ_ITEMQ;\n        for (int i = tid >> 3; i < n_items; i += 32) {",
     "        const int n_items = sh.n_items < CJ_ITEMQ ? sh.n_items : CJ_ITEMQ;\n        if (tid == 0) { sh.clk[24] += 1ull; sh.clk[25] += (unsigned long long)sh.n_items; }\n        for (int i = tid >> 3; i < n_items; i += 32) {"),
]
args = [sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "walkclk", "testpath.hip"]
for a, b in PAIRS:
    args += [a, b]
subprocess.check_call(args)
