#!/usr/bin/env python3
"""Round 6: variants of the LAST source whose k_seg_walk computed the call rows' medians at the end of its workgroup
(commit 09903b6^, wrong medians in about one call of five at 125 x 50 kb), to find what about it fails:

    python tools/walktail_variants.py            # builds wisecondor_amd/ab/lib_wt_<variant>.so for every variant
    tools/walktail_run.sh                        # (GPU box) 150 calls per variant, differences counted

The old testpath.hip is taken from git (never committed twice); the other objects are the current build's (only
testpath.hip differs between that commit and HEAD).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wisecondor_amd.build import CSRC, FLAGS, SOURCES, _hipcc  # noqa: E402

AB = os.path.join(ROOT, "wisecondor_amd", "ab")
OLD = subprocess.check_output(["git", "-C", ROOT, "show", "09903b6^:wisecondor_amd/csrc/testpath.hip"]).decode()

SELECT_OLD = """            const double *src = staged ? sv : rr;
            lo = block_select<256>(src, Ls, (Ls - 1) / 2, tid);
            hi = (Ls & 1) ? lo : block_select<256>(src, Ls, Ls / 2, tid);"""
# the same selection, but the LDS copy and the global array each through a pointer of its own address space
# (the old form hands block_select a GENERIC pointer -- `staged ? sv : rr` -- and the loads become flat_load)
SELECT_TYPED = """            if (staged) {
                lo = block_select<256>((const double *)sv, Ls, (Ls - 1) / 2, tid);
                hi = (Ls & 1) ? lo : block_select<256>((const double *)sv, Ls, Ls / 2, tid);
            } else {
                lo = block_select<256>(rr, Ls, (Ls - 1) / 2, tid);
                hi = (Ls & 1) ? lo : block_select<256>(rr, Ls, Ls / 2, tid);
            }"""
# never staged: every selection reads the ratios in place (global loads only)
SELECT_GLOBAL = """            lo = block_select<256>(rr, Ls, (Ls - 1) / 2, tid);
            hi = (Ls & 1) ? lo : block_select<256>(rr, Ls, Ls / 2, tid);"""
assert OLD.count(SELECT_OLD) == 1

VARIANTS = {
    # name: (source edits, extra flags)
    "control": ([], []),
    "typed": ([(SELECT_OLD, SELECT_TYPED)], []),
    "global": ([(SELECT_OLD, SELECT_GLOBAL)], []),
    "waitzero": ([], ["-mllvm", "-amdgpu-waitcnt-forcezero=1"]),
    # 256 VGPRs per lane instead of 128: no VGPR spill, no scratch memory at all
    "lb2": ([("__global__ __launch_bounds__(256, 4) void k_seg_walk(", "__global__ __launch_bounds__(256, 2) void k_seg_walk(")], []),
    # the exact evaluation's four wave scratches in LDS of their own instead of aliasing the search's staging area
    "ownsc": ([("    wc::PwWaveScratch *sc = reinterpret_cast<wc::PwWaveScratch *>(sh.pn);      // four of them fit pn .. itemq\n    static_assert(4 * sizeof(wc::PwWaveScratch) <= sizeof(sh.pn) + sizeof(sh.b8x) + sizeof(sh.b8n) + sizeof(sh.q2) +\n                                                       sizeof(sh.q1) + sizeof(sh.l1) + sizeof(sh.itemq),\n                  \"the exact evaluation's scratch does not fit the search's staging area\");\n    const int lane = tid & 63, w = tid >> 6;\n    int wins = 0, evals = 0;\n    if (tid == 0) {\n        Job root;",
                "    __shared__ wc::PwWaveScratch sc_own[4];\n    wc::PwWaveScratch *sc = sc_own;\n    const int lane = tid & 63, w = tid >> 6;\n    int wins = 0, evals = 0;\n    if (tid == 0) {\n        Job root;")], []),
    "O1": ([], ["-O1"]),
    "noinline": ([("template <int NT = CP_THREADS>      // NT >= 256 threads\n__device__ inline double block_select(",
                   "template <int NT = CP_THREADS>      // NT >= 256 threads\n__device__ __attribute__((noinline)) double block_select(")], []),
}

# instrumented: (1) a canary word per wave, written at the kernel's start, checked at the tail's start and end (a foreign
# write into this workgroup's LDS); (2) a step check at the loop top, the tail's start and every selection pass: each wave
# posts the step's number, a barrier, every thread compares the four -- a wave out of step prints its position
PROBE_DECL = """    __shared__ int s_canary[4], s_step[4];
#define WT_STEP(id) do { if ((tid & 63) == 0) s_step[tid >> 6] = (id); __syncthreads(); \\
        if (s_step[0] != (id) || s_step[1] != (id) || s_step[2] != (id) || s_step[3] != (id)) { \\
            if ((tid & 63) == 0) printf("WT_STEP region %d wave %d at step %d sees %d %d %d %d\\\\n", (int)blockIdx.x, tid >> 6, (id), s_step[0], s_step[1], s_step[2], s_step[3]); } \\
        __syncthreads(); } while (0)
#define WT_CANARY(where) do { if ((tid & 63) == 0 && s_canary[tid >> 6] != 0x5A000000 + (int)blockIdx.x) \\
        printf("WT_CANARY region %d wave %d at %d holds %08x\\\\n", (int)blockIdx.x, tid >> 6, (where), s_canary[tid >> 6]); } while (0)
"""
VARIANTS["probe"] = ([
    ("    const int region = blockIdx.x, tid = threadIdx.x;\n    if (region >= n_regions) return;\n    const Region rg = regions[region];\n    if (rg.n <= 0) return;                         // (out_n was zeroed by the set-up kernel)\n    if (rg.n > CJ_MAXLEN || !reg_flag[region]) {\n        if (tid == 0) counters[6] = 1;",
     PROBE_DECL + "    const int region = blockIdx.x, tid = threadIdx.x;\n    if (region >= n_regions) return;\n    if ((tid & 63) == 0) s_canary[tid >> 6] = 0x5A000000 + (int)blockIdx.x;\n    int wt_iter = 0;\n    const Region rg = regions[region];\n    if (rg.n <= 0) return;                         // (out_n was zeroed by the set-up kernel)\n    if (rg.n > CJ_MAXLEN || !reg_flag[region]) {\n        if (tid == 0) counters[6] = 1;"),
    ("        if (tid == 0) --s_sp;\n        if (job.hi - job.lo <= 0) continue;\n        const CellGeom g = cell_setup(sh, job, rg, region, prefix, rs, tmin, tmax, tmin2, tmax2, tid);\n        cell_seed(sh, g, T, tid);\n        __syncthreads();\n        double vmax = -INFINITY, vmin = INFINITY, d2 = -INFINITY, d3 = INFINITY;\n        cell_search<0>(sh, g, rs, eps2, INFINITY, -INFINITY, nullptr, 0, 1, vmax, vmin, wins, evals, tid);",
     "        if (tid == 0) --s_sp;\n        ++wt_iter;\n        WT_STEP(1000 + wt_iter);\n        if (job.hi - job.lo <= 0) continue;\n        const CellGeom g = cell_setup(sh, job, rg, region, prefix, rs, tmin, tmax, tmin2, tmax2, tid);\n        cell_seed(sh, g, T, tid);\n        __syncthreads();\n        double vmax = -INFINITY, vmin = INFINITY, d2 = -INFINITY, d3 = INFINITY;\n        cell_search<0>(sh, g, rs, eps2, INFINITY, -INFINITY, nullptr, 0, 1, vmax, vmin, wins, evals, tid);\n        WT_STEP(2000 + wt_iter);"),
    ("    const int nseg = s_nseg;\n    if (tid == 0) {\n        out_n[region] = nseg;\n        atomicAdd(&counters[4], nseg);\n    }\n    double *sv = sh.pn; ",
     "    WT_STEP(3000);\n    WT_CANARY(1);\n    const int nseg = s_nseg;\n    if (tid == 0) {\n        out_n[region] = nseg;\n        atomicAdd(&counters[4], nseg);\n    }\n    double *sv = sh.pn; "),
    ("        const bool has_nan = s_nan != 0;\n        double lo = 0.0, hi = 0.0;\n        if (!has_nan) {\n            const double *src = staged ? sv : rr;",
     "        const bool has_nan = s_nan != 0;\n        WT_STEP(4000 + sidx);\n        double lo = 0.0, hi = 0.0;\n        if (!has_nan) {\n            const double *src = staged ? sv : rr;"),
    ("            o[3] = seg_val[sidx];\n            o[4] = med - 1.0;\n        }\n    }\n    if (work) {\n        for (int o = 32; o > 0; o >>= 1) { evals += __shfl_xor(evals, o); wins += __shfl_xor(wins, o); }\n        const int slot = (int)((blockIdx.x * 7u + (unsigned)w) & 63u);",
     "            o[3] = seg_val[sidx];\n            o[4] = med - 1.0;\n        }\n        WT_STEP(5000 + sidx);\n    }\n    WT_CANARY(2);\n    if (work) {\n        for (int o = 32; o > 0; o >>= 1) { evals += __shfl_xor(evals, o); wins += __shfl_xor(wins, o); }\n        const int slot = (int)((blockIdx.x * 7u + (unsigned)w) & 63u);"),
], [])

# THE FIX (round 6): the release half of __syncthreads() -- s_waitcnt lgkmcnt(0) before s_barrier, so that this wave's LDS
# writes have completed when the others are released -- is MISSING in the compiler's output at the head of the walk
# loop (`.LBB.._12: s_barrier` straight after the back edges that carry thread 0's ds_write of s_sp / stack[]): the
# other waves can read a stale stack pointer, take a different job, and fall out of step.  An explicit wait in front of
# every barrier restores it.
VARIANTS["waitfix"] = ([("#include \"ctx.h\"", "#include \"ctx.h\"\n#define __syncthreads() do { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"workgroup\"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"workgroup\"); } while (0)")], [])


def build(name):
    edits, flags = VARIANTS[name]
    text = OLD
    for old, new in edits:
        assert text.count(old) == 1, (name, old[:60])
        text = text.replace(old, new)
    src = os.path.join(CSRC, "_wt_%s_testpath.hip" % name)
    open(src, "w").write(text)
    obj = os.path.join(AB, "wt_%s.o" % name)
    try:
        subprocess.check_call([_hipcc()] + FLAGS + flags + ["-c", src, "-o", obj])
        if "--asm" in sys.argv:
            subprocess.check_call([_hipcc()] + FLAGS + flags + ["-S", "--cuda-device-only", src, "-o",
                                                                os.path.join(AB, "wt_%s.s" % name)])
    finally:
        os.remove(src)
    objs = [obj if s == "testpath.hip" else os.path.join(CSRC, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    out = os.path.join(AB, "lib_wt_%s.so" % name)
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lz"])
    return out


if __name__ == "__main__":
    os.makedirs(AB, exist_ok=True)
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(VARIANTS)
    procs = []
    for n in names:
        print(build(n), flush=True)
