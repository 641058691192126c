#!/usr/bin/env python3
"""Busy / cache counters of the newref kernels from tools/pmc_run.sh passes -> profiles/<tag>_pmc_busy.{md,json}.

    python3 tools/busy_summary.py r02 gpurun_out/r2A_cfg2 gpurun_out/r2A_cfg4 gpurun_out/r2T_cfg4 ...

Every directory is one `rocprofv3 --pmc <set>` pass (pmc_counter_collection.csv).  Units as the
guide states them: SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over the
1024 SIMDs, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over the SIMDs, GRBM_GUI_ACTIVE cycles summed
over the 8 XCDs."""
import csv
import json
import os
import sys
from collections import defaultdict


def read_pass(folder):
    path = None
    for root, _, files in os.walk(folder):
        for f in files:
            if f.endswith("counter_collection.csv"):
                path = os.path.join(root, f)
    acc = defaultdict(lambda: defaultdict(float))
    dur = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[(k, r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {}
    for k in acc:
        d = [v for (kk, _), v in dur.items() if kk == k]
        out[k] = {"launches": len(d), "avg_ms": sum(d) / len(d) / 1e6,
                  "counters": {c: v / len(d) for c, v in acc[k].items()}}
    return out


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    tag, folders = sys.argv[1], sys.argv[2:]
    res = {}
    for folder in folders:
        run = os.path.basename(folder.rstrip("/"))
        for k, v in read_pass(folder).items():
            if v["avg_ms"] < 0.02:
                continue
            c = v["counters"]
            row = {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 4), "counters": c}
            if "GRBM_GUI_ACTIVE" in c:
                cyc = c["GRBM_GUI_ACTIVE"] / 8.0
                row["eff_clock_ghz"] = cyc / (v["avg_ms"] * 1e6)
                if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                    row["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
                if "SQ_ACTIVE_INST_VALU" in c:
                    row["valu_busy"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / cyc
                if "SQ_ACTIVE_INST_LDS" in c:
                    row["lds_busy"] = 4.0 * c["SQ_ACTIVE_INST_LDS"] / 1024.0 / cyc
                if "SQ_WAVE_CYCLES" in c:
                    row["waves_per_simd"] = 4.0 * c["SQ_WAVE_CYCLES"] / 1024.0 / cyc
            if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
                row["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
                row["l2_miss_bytes_128B"] = c["TCC_MISS_sum"] * 128.0
            if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
                row["wait_any_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
            res.setdefault(run, {})[short(k)] = row
    os.makedirs("profiles", exist_ok=True)
    json.dump(res, open("profiles/%s_pmc_busy.json" % tag, "w"), indent=1, sort_keys=True)
    with open("profiles/%s_pmc_busy.md" % tag, "w") as f:
        f.write("# %s busy / cache counters (rocprofv3 --pmc, one pass per counter set, `tools/pmc_run.sh`, summarised by "
                "`tools/busy_summary.py`)\n\n" % tag)
        f.write("Per-launch averages of `tools/gpu_newref_only.py <cfg>` (kernel-level synthetic matrix).  Derived columns: "
                "effective clock = GRBM_GUI_ACTIVE / 8 / duration; matrix-core busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / cycles; "
                "VALU (LDS) busy = 4 x SQ_ACTIVE_INST_VALU (LDS) / 1024 / cycles; resident waves per SIMD = 4 x SQ_WAVE_CYCLES / "
                "1024 / cycles; L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS).\n\n")
        f.write("| run | kernel | launches | avg ms | derived | raw counters |\n|---|---|---|---|---|---|\n")
        for run in sorted(res):
            for k, row in sorted(res[run].items(), key=lambda kv: -kv[1]["avg_ms"]):
                derived = ", ".join("%s=%.3g" % (n, row[n]) for n in ("eff_clock_ghz", "mfma_busy", "valu_busy", "lds_busy",
                                                                       "waves_per_simd", "l2_hit_rate", "l2_miss_bytes_128B",
                                                                       "wait_any_frac") if n in row)
                raw = ", ".join("%s=%.4g" % kv for kv in sorted(row["counters"].items()))
                f.write("| %s | `%s` | %d | %.3f | %s | %s |\n" % (run, k, row["launches"], row["avg_ms"], derived, raw))
    print(json.dumps({r: {k: {n: v for n, v in row.items() if n != "counters"} for k, row in res[r].items()} for r in res},
                     indent=1))


if __name__ == "__main__":
    main()
