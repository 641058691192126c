#!/usr/bin/env python3
"""Development aid: a variant of the library whose k_seg_bound accumulates s_memtime deltas of its phases
(thread 0 of every workgroup, atomics on g_dbg[0..15]) -> wisecondor_amd/ab/lib_phase.so; read with
tools/gpu_phase_bound.py on the GPU box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [
    ("            if (cut_lo < in_lo) atomicMax(&cuts[2 * j + 1], wc::f64_ordered(-cut_lo));\n        }\n    }\n}\n",
     "            if (cut_lo < in_lo) atomicMax(&cuts[2 * j + 1], wc::f64_ordered(-cut_lo));\n        }\n    }\n    __syncthreads();\n    if (tid < 11) atomicAdd(&g_dbg[tid + 16 * ((blockIdx.y * 5 + blockIdx.x) & 3)], s_ph[tid]);\n}\n"),
    # clock helper
    ("template <int MODE, class F>\n__device__ inline void bscan_chunk(",
     "__shared__ unsigned long long s_ph[16];\n#define PH(n) do { if (MODE == 0 && tid == 0) { const unsigned long long t_ = clock64(); s_ph[n] += t_ - t_prev; t_prev = t_; } } while (0)\n"
     "template <int MODE, class F>\n__device__ inline void bscan_chunk("),
    ("    __syncthreads();                              // the previous chunk's shared state is done with\n",
     "    __syncthreads();                              // the previous chunk's shared state is done with\n"
     "    unsigned long long t_prev = clock64();\n"),
    ("        __syncthreads();\n        for (int side = 0; side < 2; ++side) {\n            int xr = chunk * ROWS_HALF + lane;\n            bool live;\n            if (side == 0) {\n                live = xr < half;\n            } else {\n                xr = L - 1 - xr;\n                live = xr >= half;\n            }\n            const long long ax = base + (live ? xr : 0);\n            const int xr_lo = side == 0 ? chunk * ROWS_HALF : L - 1 - (chunk * ROWS_HALF + 63);\n            const long long a0 = base + (xr_lo < 0 ? 0 : xr_lo);\n            const int xo = (int)(ax - a0);                              // 0..63 for live lanes\n            const double px = live ? s_pn[side][xo] : 0.0;\n            if (w == 0) { s_px",
     "        __syncthreads();\n        PH(0);\n        for (int side = 0; side < 2; ++side) {\n            int xr = chunk * ROWS_HALF + lane;\n            bool live;\n            if (side == 0) {\n                live = xr < half;\n            } else {\n                xr = L - 1 - xr;\n                live = xr >= half;\n            }\n            const long long ax = base + (live ? xr : 0);\n            const int xr_lo = side == 0 ? chunk * ROWS_HALF : L - 1 - (chunk * ROWS_HALF + 63);\n            const long long a0 = base + (xr_lo < 0 ? 0 : xr_lo);\n            const int xo = (int)(ax - a0);                              // 0..63 for live lanes\n            const double px = live ? s_pn[side][xo] : 0.0;\n            if (w == 0) { s_px"),
    ("            for (int b = fb + w; b < 16; b += 4) {\n                const int minlen = 8 * b - xo;                          // 9..\n",
     "            PH(1);\n            for (int b = fb + w; b < 16; b += 4) {\n                const int minlen = 8 * b - xo;                          // 9..\n"),
    ("            const long long org = (k_base / 4) * QB2;\n", "            PH(2);\n            const long long org = (k_base / 4) * QB2;\n"),
    ("            for (int r2 = (ra0 + FAR2) / QB2 + w; r2 <= r2_last; r2 += 4) {\n",
     "            PH(3);\n            for (int r2 = (ra0 + FAR2) / QB2 + w; r2 <= r2_last; r2 += 4) {\n"),
    ("                if ((++since & 7) == 0) refresh();\n            }\n        }\n        __syncthreads();\n        const int nwork = s_nwork;\n",
     "                if ((++since & 7) == 0) refresh();\n            }\n            PH(4);\n        }\n        __syncthreads();\n        PH(5);\n        const int nwork = s_nwork;\n"),
    ("    if (full) {\n        // every window of the block (jobs longer than the staged tables cover, queue overflow): the plain scan\n",
     "    PH(6);\n    if (full) {\n        // every window of the block (jobs longer than the staged tables cover, queue overflow): the plain scan\n"),
    ("        block_minmax4(vmax, vmin, ubmax, lbmin, tid);\n        if (tid == 0) {\n            Extreme e;\n",
     "        { const unsigned long long t0_ = clock64(); block_minmax4(vmax, vmin, ubmax, lbmin, tid); if (tid == 0) s_ph[7] += clock64() - t0_; }\n        if (tid == 0) {\n            s_ph[9] += 1ull;\n            Extreme e;\n"),
    ("    const bool table_ok = stage_block_tables(base, L, tmin, tmax, s_tmx, s_tmn, k_base, k_last, tid, tmin2, tmax2, s_tmx2, s_tmn2);\n    for (int chunk = blockIdx.x; chunk * ROWS_HALF < half; chunk += gridDim.x) {\n        // the job's cuts as every workgroup",
     "    if (tid < 16) s_ph[tid] = 0ull;\n    __syncthreads();\n    const unsigned long long tk0_ = clock64();\n    const bool table_ok = stage_block_tables(base, L, tmin, tmax, s_tmx, s_tmn, k_base, k_last, tid, tmin2, tmax2, s_tmx2, s_tmn2);\n    if (tid == 0) { s_ph[8] += clock64() - tk0_; s_ph[10] += 1ull; }\n    for (int chunk = blockIdx.x; chunk * ROWS_HALF < half; chunk += gridDim.x) {\n        // the job's cuts as every workgroup"),
]
args = [sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "phase", "testpath.hip"]
for a, b in PAIRS:
    args += [a, b]
subprocess.check_call(args)
